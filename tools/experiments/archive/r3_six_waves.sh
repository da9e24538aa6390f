# Round 3: six waves per SIMD for filter_kernel (80 VGPRs; needs the cell statistics in LDS cut to the 64 blocks of cells up to
# 64 x 64 so that six workgroups share a CU's LDS), A/B against the default five, in one call.
cd $GRAFT_REPO_ROOT
bash tools/build_variant.sh w6 -DMM_FILTER_WAVES=6 -DMM_STAT_BLOCKS=64 > /dev/null 2>&1
bash tools/build_variant.sh w5s -DMM_STAT_BLOCKS=64 > /dev/null 2>&1
for round in 1 2; do
for v in default w6 w5s; do
  if [ $v = default ]; then unset MOSAIC_HIP_LIB; else export MOSAIC_HIP_LIB=$PWD/gpurun_out/$v.so; fi
  for c in D E8 C; do
    k=6; [ $c = C ] && k=60
    python3 bench.py --config $c --steps $k --warmup 3 --no-cpu --opt phase_timing=1 2>/dev/null | python3 -c "import json,sys; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); t=d['last_step_device_ms']; print(sys.argv[1], sys.argv[2], '%.0f cells/s %.3f ms/step filter %.3f' % (d['value'], d['ms_per_step'], t['score_ms']), d.get('result_digest','')[:10])" $c $v
  done
done
done
