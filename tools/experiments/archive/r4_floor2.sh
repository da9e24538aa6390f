# Round 4: the fp16 form with the chroma floor as one integer maximum and the band at 0.29 + 6e-3 v: sweeps, maximiser, the
# goldens and full-size properties, then the step against the form of the previous commit (no floor, 0.38 + 8e-3 v) on one box.
set -x
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_floor2; mkdir -p $O
timeout -k 10 600 python3 -m pytest tests/test_gpu_filter.py -q -s > $O/pytest.log 2>&1; echo "pytest rc $?"; grep -a "sweep\|maximiser\|passed\|failed" $O/pytest.log | cut -c1-250
timeout -k 10 600 python3 -m pytest tests/test_golden_grids.py tests/test_gpu_fullsize.py -x -q -m gpu > $O/pytest2.log 2>&1; echo "pytest2 rc $?"; tail -3 $O/pytest2.log
bash tools/build_variant.sh floor0 -DMM_PRE_CHROMA_FLOOR=0 -DMM_PRE_ABS=0.38f -DMM_PRE_REL=8e-3f > $O/b1.log 2>&1
for rep in 1 2 3; do
  for c in D E8 G I; do
    for lib in gpurun_out/floor0.so ""; do
      MOSAIC_HIP_LIB=$lib python3 bench.py --config $c --steps 8 --warmup 2 --no-cpu --opt pretier=1 > $O/x.json 2>/dev/null; python3 tools/experiments/r4_line.py $O/x.json "$c pretier=1 lib=[$lib]"
    done
  done
done
