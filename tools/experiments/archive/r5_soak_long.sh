# Round 5, long soak on the final sources: 1500 fuzz scenarios on the defaults, 1500 with the pre-tier forced on and every drop audited.
set -x
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_soak_long; mkdir -p $O
export MOSAIC_FUZZ_LARGE=300 MOSAIC_FUZZ_SMALL=1000 MOSAIC_FUZZ_SMOOTH=200
timeout -k 10 560 python3 -m pytest tests/test_gpu_fuzz.py -x -q -p no:cacheprovider > $O/soak_default.log 2>&1; echo "rc $?"; tail -2 $O/soak_default.log
MOSAIC_TEST_OPTS="pretier=1,audit=1" timeout -k 10 560 python3 -m pytest tests/test_gpu_fuzz.py -x -q -p no:cacheprovider > $O/soak_pre_audit.log 2>&1; echo "rc $?"; tail -2 $O/soak_pre_audit.log
