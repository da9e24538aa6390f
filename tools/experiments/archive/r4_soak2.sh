# Round 4: the soak once more on the final sources, now ASSERTING that no launch was voided: with audit = 1 every candidate any
# scenario dropped is walked again without the bound.
set -x
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_final; mkdir -p $O
MOSAIC_TEST_OPTS="pretier=1,audit=1" MOSAIC_FUZZ_LARGE=300 MOSAIC_FUZZ_SMALL=600 MOSAIC_FUZZ_SMOOTH=300 timeout -k 10 800 python3 -m pytest tests/test_gpu_fuzz.py -x -q -p no:cacheprovider -k "not tuning_options" > $O/soak_audit_all.log 2>&1; echo "rc $?"; tail -2 $O/soak_audit_all.log
