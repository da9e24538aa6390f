# Round 4, last pass: the whole GPU suite on the final sources, then the soak of the fuzz families with every drop audited.
set -x
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_final; mkdir -p $O
timeout -k 10 600 python3 -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log; tail -4 $O/pytest.log | cut -c1-300
MOSAIC_TEST_OPTS="pretier=1,audit=1" MOSAIC_FUZZ_LARGE=200 MOSAIC_FUZZ_SMALL=600 MOSAIC_FUZZ_SMOOTH=200 timeout -k 10 500 python3 -m pytest tests/test_gpu_fuzz.py -x -q -p no:cacheprovider -k "not tuning_options" > $O/soak_audit_all.log 2>&1; echo "rc $?"; tail -2 $O/soak_audit_all.log
