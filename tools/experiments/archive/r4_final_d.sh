# Round 4, last pass: the default-options soak, then the PMC passes of ten configurations on the final kernels.
set -x
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_final; mkdir -p $O
MOSAIC_FUZZ_LARGE=200 MOSAIC_FUZZ_SMALL=600 MOSAIC_FUZZ_SMOOTH=200 timeout -k 10 540 python3 -m pytest tests/test_gpu_fuzz.py -x -q -p no:cacheprovider > $O/soak_default.log 2>&1; echo "rc $?"; tail -2 $O/soak_default.log
CONFIGS="D C B E8 A F G H I P" ONLY=pmc bash tools/profile.sh r04_v46
