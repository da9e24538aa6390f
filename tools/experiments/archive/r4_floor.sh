# Round 4: what the chroma floor of the fp16 form costs per step (same box, alternating), and what a band narrowed to twice
# the new worst swept pair gives back.
set -x
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_floor; mkdir -p $O
bash tools/build_variant.sh floor0 -DMM_PRE_CHROMA_FLOOR=0 > $O/b1.log 2>&1
bash tools/build_variant.sh band26 -DMM_PRE_ABS=0.26f -DMM_PRE_REL=5.5e-3f > $O/b2.log 2>&1
bash tools/build_variant.sh band26r8 -DMM_PRE_ABS=0.26f -DMM_PRE_REL=8e-3f > $O/b3.log 2>&1
MOSAIC_HIP_LIB=gpurun_out/band26.so MM_TEST_PRE_BAND=0.26,5.5e-3 timeout -k 10 600 python3 -m pytest tests/test_gpu_filter.py -q -s -k "pretier_band or maximiser" > $O/pytest26.log 2>&1; echo "pytest rc $?"; grep -a "sweep\|maximiser\|passed\|failed" $O/pytest26.log | cut -c1-250
MOSAIC_HIP_LIB=gpurun_out/band26r8.so MM_TEST_PRE_BAND=0.26,8e-3 timeout -k 10 600 python3 -m pytest tests/test_gpu_filter.py -q -s -k "pretier_band or maximiser" > $O/pytest26r8.log 2>&1; echo "pytest rc $?"; grep -a "sweep\|maximiser\|passed\|failed" $O/pytest26r8.log | cut -c1-250
for rep in 1 2 3; do
  for c in D E8 G; do
    for lib in gpurun_out/floor0.so "" gpurun_out/band26.so gpurun_out/band26r8.so; do
      MOSAIC_HIP_LIB=$lib python3 bench.py --config $c --steps 8 --warmup 2 --no-cpu --opt pretier=1 > $O/x.json 2>/dev/null; python3 tools/experiments/r4_line.py $O/x.json "$c pretier=1 lib=[$lib]"
    done
  done
done
