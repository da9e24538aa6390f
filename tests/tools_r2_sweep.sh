set -x
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-r2b}
mkdir -p $O
python -m pytest tests/test_gpu_filter.py tests/test_gpu_generator.py tests/test_gpu_fullsize.py -m gpu -x -q > $O/tests.log 2>&1 || { tail -30 $O/tests.log; exit 1; }
tail -3 $O/tests.log
B="python3 bench.py --steps 6 --warmup 2 --no-cpu"
g() { grep -o "\"value\": [0-9.]*\|\"score_ms\": [0-9.]*\|\"visited_groups\": [0-9]*\|\"bound_dropped\": [0-9]*" | tr '\n' ' '; echo; }
echo "== default"; $B 2>$O/err.log | tee $O/bench_default.json | g
echo "== xcd_aware=0"; $B --opt xcd_aware=0 2>/dev/null | g
for c in 16 32 128; do echo "== xcd_cells=$c"; $B --opt xcd_cells=$c 2>/dev/null | g; done
for t in 16 64; do echo "== tile=$t"; $B --opt tile=$t 2>/dev/null | g; done
tests/tools_build_variant.sh w4 -DMM_FILTER_WAVES=4 && { echo "== waves 4"; MOSAIC_HIP_LIB=gpurun_out/w4.so $B 2>/dev/null | g; }
tests/tools_build_variant.sh w6 -DMM_FILTER_WAVES=6 && { echo "== waves 6"; MOSAIC_HIP_LIB=gpurun_out/w6.so $B 2>/dev/null | g; }
rm -f gpurun_out/w4.so gpurun_out/w6.so
