# Round 5: where a wave's cycles go inside filter_kernel's items on the final sources (instrumented build, -DMM_FILTER_STAMPS).
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
MM_ENGINE_FLAGS=-DMM_FILTER_STAMPS bash tools/build_variant.sh stamps -DMM_FILTER_STAMPS > /dev/null 2>&1
O=gpurun_out/r05_stamps.txt; : > $O
for c in ${CONFIGS:-D G E8}; do
  echo "== $c" >> $O
  MOSAIC_HIP_LIB=$PWD/gpurun_out/stamps.so python3 tools/visit_counters.py $c fast_path=0 2>&1 | grep -v amdgpu.ids >> $O
done
cat $O
