# Round 5 soak: the fuzz scenarios with the pre-tier FORCED on and EVERY drop audited, on the worst-case band and on the observed one; then the defaults.
set -x
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_soak; mkdir -p $O
export MOSAIC_FUZZ_LARGE=120 MOSAIC_FUZZ_SMALL=300 MOSAIC_FUZZ_SMOOTH=100
MOSAIC_TEST_OPTS="pretier=1,audit=1" timeout -k 10 500 python3 -m pytest tests/test_gpu_fuzz.py -x -q -p no:cacheprovider > $O/soak_pre_audit.log 2>&1; echo "rc $?"; tail -2 $O/soak_pre_audit.log
MOSAIC_TEST_OPTS="pretier=1,pretier_band=1,audit=1" timeout -k 10 500 python3 -m pytest tests/test_gpu_fuzz.py -x -q -p no:cacheprovider > $O/soak_pre_observed_audit.log 2>&1; echo "rc $?"; tail -2 $O/soak_pre_observed_audit.log
timeout -k 10 500 python3 -m pytest tests/test_gpu_fuzz.py -x -q -p no:cacheprovider > $O/soak_default.log 2>&1; echo "rc $?"; tail -2 $O/soak_default.log
