# Round 4: what the pre-tier is worth at three band widths (share of the band the worst swept pair uses: 0.99 / 0.72 / 0.50).
set -x
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_pre4; mkdir -p $O
bash tools/build_variant.sh band16 -DMM_PRE_ABS=0.16f -DMM_PRE_REL=8e-3f > $O/b1.log 2>&1
bash tools/build_variant.sh band35 -DMM_PRE_ABS=0.35f -DMM_PRE_REL=8e-3f > $O/b2.log 2>&1
for rep in 1 2; do
  for c in D E8 G I C; do
    python3 bench.py --config $c --steps 8 --warmup 2 --no-cpu --opt pretier=0 > $O/x.json 2>/dev/null; python3 tools/experiments/r4_line.py $O/x.json "$c pretier=0"
    for lib in "" gpurun_out/band16.so gpurun_out/band35.so; do
      MOSAIC_HIP_LIB=$lib python3 bench.py --config $c --steps 8 --warmup 2 --no-cpu --opt pretier=1 > $O/x.json 2>/dev/null; python3 tools/experiments/r4_line.py $O/x.json "$c pretier=1 lib=[$lib]"
    done
  done
done
