# Round 4: the full GPU suite on the committed fp16 form (chroma floor, band 0.29 + 8e-3 v), then the long maximiser on it.
set -x
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_maxlong; mkdir -p $O
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?"; tail -4 $O/pytest.log
timeout -k 10 400 python3 tools/experiments/r4_maximiser_long.py 8 400 > $O/long.txt 2>&1; echo "rc $?"; cut -c1-230 $O/long.txt
