# A/B of the working tree's filter.hip against a copy of an earlier one (csrc/filter_prev.hip, not committed), same box.
cd $GRAFT_REPO_ROOT
C=mosaicmagnifique_amd/csrc
mkdir -p gpurun_out/var_prev
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wextra -Iinclude"
hipcc $F -ffp-contract=off -c $C/engine.hip -o gpurun_out/var_prev/engine.o 2>/dev/null
hipcc $F -ffp-contract=fast -fno-slp-vectorize -c $C/filter_prev.hip -o gpurun_out/var_prev/filter.o 2>/dev/null
hipcc $F -c $C/group.hip -o gpurun_out/var_prev/group.o 2>/dev/null
hipcc --offload-arch=gfx950 -shared -fPIC -o gpurun_out/prev.so gpurun_out/var_prev/engine.o gpurun_out/var_prev/filter.o gpurun_out/var_prev/group.o -L/opt/rocm/lib -lrccl -Wl,-rpath,/opt/rocm/lib -lpthread
for round in 1 2 3; do
for v in default prev; do
  if [ $v = default ]; then unset MOSAIC_HIP_LIB; else export MOSAIC_HIP_LIB=$PWD/gpurun_out/prev.so; fi
  python3 bench.py --config D --steps 6 --warmup 3 --no-cpu 2>/dev/null | python3 -c "import json,sys; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); t=d['last_step_device_ms']; print('D', sys.argv[1], '%.0f cells/s %.3f ms/step filter %.3f' % (d['value'], d['ms_per_step'], t['score_ms']), d.get('result_digest','')[:10])" $v
done
done
