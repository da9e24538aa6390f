# Round 5, soak over scenarios no earlier run has seen (MOSAIC_FUZZ_BASE: the seeds of every fuzz family start there): the defaults, then the
# pre-tier forced on with every drop audited.
set -x
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_soak_fresh; mkdir -p $O
export MOSAIC_FUZZ_BASE=${BASE:-100000} MOSAIC_FUZZ_LARGE=300 MOSAIC_FUZZ_SMALL=1000 MOSAIC_FUZZ_SMOOTH=200 MOSAIC_FUZZ_STEPS=120
timeout -k 10 570 python3 -m pytest tests/test_gpu_fuzz.py -x -q -p no:cacheprovider > $O/soak_default.log 2>&1; echo "rc $?"; tail -3 $O/soak_default.log
MOSAIC_TEST_OPTS="pretier=1,audit=1" timeout -k 10 570 python3 -m pytest tests/test_gpu_fuzz.py -x -q -p no:cacheprovider > $O/soak_pre_audit.log 2>&1; echo "rc $?"; tail -3 $O/soak_pre_audit.log
