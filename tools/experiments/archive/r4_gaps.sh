# Round 4: what a HIP graph could buy the small jobs -- the device-side gaps between the kernels of a step (rocprofv3 kernel trace of
# configs H, C, B; tools/experiments/r4_gaps.py sums, per step, the kernels' durations and the idle time between them).
set -x
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_gaps; mkdir -p $O
for c in H C B; do
  rocprofv3 --kernel-trace -d $O/kt_$c -o t --output-format csv -- python3 bench.py --config $c --steps 20 --warmup 3 --no-cpu --no-floors > $O/bench_$c.json 2> $O/kt_$c.log
done
python3 tools/experiments/r4_gaps.py $O > $O/gaps.txt; cat $O/gaps.txt
