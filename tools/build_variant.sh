#!/bin/bash
# Builds a variant of the engine with extra -D flags for filter.hip into gpurun_out/<name>.so
# (MM_ENGINE_FLAGS: extra flags for engine.hip)
# usage: tools/build_variant.sh NAME -DMM_FILTER_WAVES=4 ...   then MOSAIC_HIP_LIB=gpurun_out/NAME.so python bench.py
set -e
name=$1; shift
C=mosaicmagnifique_amd/csrc
mkdir -p gpurun_out/var_$name
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wextra -Iinclude"
hipcc $F -ffp-contract=off $MM_ENGINE_FLAGS -c $C/engine.hip -o gpurun_out/var_$name/engine.o
hipcc $F -ffp-contract=fast -fno-slp-vectorize "$@" -c $C/filter.hip -o gpurun_out/var_$name/filter.o
hipcc $F -c $C/group.hip -o gpurun_out/var_$name/group.o
hipcc --offload-arch=gfx950 -shared -fPIC -o gpurun_out/$name.so gpurun_out/var_$name/engine.o gpurun_out/var_$name/filter.o gpurun_out/var_$name/group.o -L${ROCM_PATH:-/opt/rocm}/lib -lrccl -Wl,-rpath,${ROCM_PATH:-/opt/rocm}/lib -lpthread
rm -rf gpurun_out/var_$name
echo gpurun_out/$name.so
