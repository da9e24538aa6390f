# Round profile of the default bench command on the GPU box: bench lines of every config, the
# rocprofv3 kernel trace, and the PMC passes (separate runs) that bench.py's roofline is priced from.
# usage (on the box): bash tests/tools_profile.sh r02_v12        -> gpurun_out/r02_v12/
set -x
V=${1:-r02}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/$V
mkdir -p $O
rocprofv3 --pmc FETCH_SIZE -d $O/fetch -o p --output-format csv -- python3 bench.py --config D --steps 1 --warmup 0 --no-cpu > $O/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $O/write -o p --output-format csv -- python3 bench.py --config D --steps 1 --warmup 0 --no-cpu > $O/write.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES GRBM_GUI_ACTIVE -d $O/sq -o p --output-format csv -- python3 bench.py --config D --steps 1 --warmup 0 --no-cpu > $O/bench_sq.json 2> $O/sq.log
python3 profiles/summarise_pmc.py D $O > $O/pmc_summary.json
cp profiles/pmc_traffic.json $O/pmc_traffic.json
rocprofv3 --kernel-trace --stats -d $O/kt -o d --output-format csv -- python3 bench.py --steps 3 --warmup 1 --no-cpu > $O/kt.log 2>&1
python3 bench.py --steps 5 --warmup 2 > $O/bench_D.json 2> $O/bench_D.err
for c in A B C E8; do python3 bench.py --config $c --steps 5 --warmup 2 --no-cpu > $O/bench_$c.json 2>/dev/null; done
MOSAIC_BENCH_FORCE_DIST=1 python3 bench.py --steps 5 --warmup 2 --no-cpu > $O/bench_D_dist1.json 2> $O/bench_D_dist1.err
find $O -name "*.csv" | head -20
