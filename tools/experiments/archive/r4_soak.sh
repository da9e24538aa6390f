# Round 4: soak of the final kernels -- the fuzz families with many more seeds than the suite holds, once on the shipped defaults
# and once with the packed-fp16 pre-tier forced on every CIEDE2000 launch (it is not eligible at these sizes by itself) and
# EVERY candidate it drops audited by the fp32 form.
set -x
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_soak; mkdir -p $O
MOSAIC_TEST_OPTS="pretier=1,audit=1" MOSAIC_FUZZ_LARGE=200 MOSAIC_FUZZ_SMALL=600 MOSAIC_FUZZ_SMOOTH=200 timeout -k 10 540 python3 -m pytest tests/test_gpu_fuzz.py -x -q -p no:cacheprovider -k "not tuning_options" > $O/soak_pretier.log 2>&1; echo "rc $?"; tail -3 $O/soak_pretier.log
MOSAIC_FUZZ_LARGE=200 MOSAIC_FUZZ_SMALL=600 MOSAIC_FUZZ_SMOOTH=200 timeout -k 10 540 python3 -m pytest tests/test_gpu_fuzz.py -x -q -p no:cacheprovider > $O/soak_default.log 2>&1; echo "rc $?"; tail -3 $O/soak_default.log
