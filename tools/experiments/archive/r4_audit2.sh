set -x
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_audit; mkdir -p $O
timeout -k 10 300 python3 -m pytest tests/test_gpu_filter.py -x -q -s -k "audit" > $O/pytest.log 2>&1; echo "pytest rc $?"; grep -a "broken\|passed\|failed\|Error\|assert" $O/pytest.log | cut -c1-250
