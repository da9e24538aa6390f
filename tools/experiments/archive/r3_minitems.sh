# Round 3: how many workgroups a small launch should start (filter_lds_kernel's cap; filter_kernel's rule is in launch_filter).
cd $GRAFT_REPO_ROOT
for m in 512 256 1024; do bash tools/build_variant.sh cap$m -DMM_LDS_GRID_CAP=$m > /dev/null 2>&1; done
for round in 1 2; do
for v in default cap1024 cap512 cap256; do
  if [ $v = default ]; then unset MOSAIC_HIP_LIB; else export MOSAIC_HIP_LIB=$PWD/gpurun_out/$v.so; fi
  for c in B A F; do
  k=200; [ $c = F ] && k=50
  python3 bench.py --config $c --steps $k --warmup 3 --no-cpu --opt phase_timing=1 2>/dev/null | python3 -c "import json,sys; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); t=d['last_step_device_ms']; print(sys.argv[2], sys.argv[1], '%.0f cells/s %.4f ms/step filter %.3f' % (d['value'], d['ms_per_step'], t['score_ms']))" $v $c
  done
done
done
python3 bench.py --config H --steps 200 --warmup 3 --no-cpu --opt phase_timing=1 2>/dev/null | python3 -c "import json,sys; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); t=d['last_step_device_ms']; print('H default', '%.0f cells/s %.4f ms/step filter %.3f' % (d['value'], d['ms_per_step'], t['score_ms']))"
