# A/B of the working tree's filter.hip against an earlier one (tests/_prev/filter.hip: scratch, not committed), same box.
cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/csrc_prev; mkdir -p gpurun_out/csrc_prev gpurun_out/var_prev
cp mosaicmagnifique_amd/csrc/*.h* mosaicmagnifique_amd/csrc/*.hip gpurun_out/csrc_prev/ 2>/dev/null
cp tests/_prev/* gpurun_out/csrc_prev/
C=gpurun_out/csrc_prev
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wextra -Iinclude"
hipcc $F -ffp-contract=off -c $C/engine.hip -o gpurun_out/var_prev/engine.o 2>/dev/null
hipcc $F -ffp-contract=fast -fno-slp-vectorize -c $C/filter.hip -o gpurun_out/var_prev/filter.o 2>/dev/null
hipcc $F -c $C/group.hip -o gpurun_out/var_prev/group.o 2>/dev/null
hipcc --offload-arch=gfx950 -shared -fPIC -o gpurun_out/prev.so gpurun_out/var_prev/engine.o gpurun_out/var_prev/filter.o gpurun_out/var_prev/group.o -L/opt/rocm/lib -lrccl -Wl,-rpath,/opt/rocm/lib -lpthread
for round in 1 2 3; do
for v in default prev; do
  if [ $v = default ]; then unset MOSAIC_HIP_LIB; else export MOSAIC_HIP_LIB=$PWD/gpurun_out/prev.so; fi
  for c in ${CONFIGS:-D E8 C H}; do
    k=6; [ $c = C ] && k=60; [ $c = H ] && k=200
    python3 bench.py --config $c --steps $k --warmup 3 --no-cpu --opt phase_timing=1 2>/dev/null | python3 -c "import json,sys; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); t=d['last_step_device_ms']; print(sys.argv[1], sys.argv[2], '%.0f cells/s %.4f ms/step filter %.3f seed %.3f' % (d['value'], d['ms_per_step'], t['score_ms'], t['seed_ms']), d.get('result_digest','')[:10])" $c $v
  done
done
done
