# Round 3: workgroups of seed_score_kernel (grid-stride; 3 waves per SIMD = 768 resident workgroups) -- cap 8192 against fewer.
cd $GRAFT_REPO_ROOT
for m in 768 1536 3072; do bash tools/build_variant.sh sg$m -DMM_SEED_GRID_CAP=$m > /dev/null 2>&1; done
for round in 1 2; do
for v in default sg768 sg1536 sg3072; do
  if [ $v = default ]; then unset MOSAIC_HIP_LIB; else export MOSAIC_HIP_LIB=$PWD/gpurun_out/$v.so; fi
  for c in D C H E8; do
  k=6; [ $c = C ] && k=60; [ $c = H ] && k=200
  python3 bench.py --config $c --steps $k --warmup 3 --no-cpu --opt phase_timing=1 2>/dev/null | python3 -c "import json,sys; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); t=d['last_step_device_ms']; print(sys.argv[2], sys.argv[1], '%.0f cells/s %.4f ms/step seed %.3f' % (d['value'], d['ms_per_step'], t['seed_ms']))" $v $c
  done
done
done
