# Round 3: C' = v_sqrt_f32(m) instead of m * v_rsq_f32(m) in the CIEDE2000 filter form (8 multiplies fewer per 4 pixel pairs),
# A/B against a build with -DMM_CHROMA_RSQ in one call.
cd $GRAFT_REPO_ROOT
bash tools/build_variant.sh rsq -DMM_CHROMA_RSQ > /dev/null 2>&1
for round in 1 2 3; do
for v in default rsq; do
  if [ $v = default ]; then unset MOSAIC_HIP_LIB; else export MOSAIC_HIP_LIB=$PWD/gpurun_out/$v.so; fi
  for c in D E8 C; do
    k=6; [ $c = C ] && k=60
    python3 bench.py --config $c --steps $k --warmup 3 --no-cpu --opt phase_timing=1 2>/dev/null | python3 -c "import json,sys; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); t=d['last_step_device_ms']; print(sys.argv[1], sys.argv[2], '%.0f cells/s %.3f ms/step filter %.3f seed %.3f' % (d['value'], d['ms_per_step'], t['score_ms'], t['seed_ms']), d.get('result_digest','')[:10])" $c $v
  done
done
done
