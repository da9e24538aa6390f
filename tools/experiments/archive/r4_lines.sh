# The bench lines of every configuration once more (after bench.py's last changes; the kernel sources are the profiled ones).
set -x
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_v48; mkdir -p $O
python3 bench.py --steps 10 --warmup 2 > $O/bench_D.json 2> $O/bench_D.err
for c in A B C E8 F G H I P; do python3 bench.py --config $c --steps 5 --warmup 2 > $O/bench_$c.json 2>/dev/null; done
python3 bench.py --config E --steps 2 --warmup 1 --no-cpu > $O/bench_E_1gpu.json 2>/dev/null
MOSAIC_BENCH_FORCE_DIST=1 python3 bench.py --steps 5 --warmup 2 --no-cpu > $O/bench_D_dist1.json 2> $O/bench_D_dist1.err
ls -la $O/bench_*.json | wc -l
