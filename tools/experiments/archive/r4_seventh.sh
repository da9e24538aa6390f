# Round 4, seventh GPU call: the tests added after the fifth call (sharded group against the goldens), the C++ host.
set -x
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_seventh; mkdir -p $O
timeout -k 10 900 python3 -m pytest tests/test_golden_grids.py tests/test_gpu_host_cpp.py tests/test_gpu_group.py tests/test_gpu_bench_ranks.py -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log; tail -6 $O/pytest.log | cut -c1-300
