# Round profile of the bench commands on the GPU box: the rocprofv3 PMC passes (separate runs, as the
# MI355X guide prescribes) that bench.py's roofline is priced from -- for configs D, C, B, E8, E, A, F, G, H, I, P and PD
# (CONFIGS="E8 E" bash tools/profile.sh ... for a subset; ONLY=pmc skips the bench lines) --, the
# kernel trace of the default command, and the bench lines of every configuration.
# usage (on the box): bash tools/profile.sh r02_v13        -> gpurun_out/r02_v13/
set -x
V=${1:-r06_v61}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/$V
mkdir -p $O
for c in ${CONFIGS:-D C B E8 E A F G H I P PD}; do
  P=$O/pmc_$c
  mkdir -p $P
  CMD="python3 bench.py --config $c --steps 1 --warmup 3 --no-cpu"      # (three warm-up steps: a job eligible for the fp16 pre-tier spends a warm-up launch and one launch each way on measuring it; the counted one runs the form it kept)
  rocprofv3 --pmc FETCH_SIZE -d $P/fetch -o p --output-format csv -- $CMD > $P/fetch.log 2>&1
  rocprofv3 --pmc WRITE_SIZE -d $P/write -o p --output-format csv -- $CMD > $P/write.log 2>&1
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES GRBM_GUI_ACTIVE -d $P/sq -o p --output-format csv -- $CMD > $P/bench_sq.json 2> $P/sq.log
  python3 profiles/summarise_pmc.py $c $P > $P/pmc_summary.json
done
cp profiles/pmc_traffic.json $O/pmc_traffic.json
[ "$ONLY" = pmc ] && exit 0
rocprofv3 --kernel-trace --stats -d $O/kt -o d --output-format csv -- python3 bench.py --steps 3 --warmup 1 --no-cpu > $O/kt.log 2>&1
python3 bench.py --steps 20 --warmup 5 > $O/bench_D.json 2> $O/bench_D.err      # (the driver's own command: the line carries every BASELINE configuration)
for c in A B C E8 F G H I P PD; do python3 bench.py --config $c --steps 5 --warmup 6 > $O/bench_$c.json 2>/dev/null; done      # (six warm-up steps: with a repeat penalty the engine tries list lengths over its first four generates)
rocprofv3 --kernel-trace --stats -d $O/kt_PD -o d --output-format csv -- python3 bench.py --config PD --steps 3 --warmup 6 --no-cpu > $O/kt_PD.log 2>&1
python3 bench.py --config E --steps 2 --warmup 1 --no-cpu > $O/bench_E_1gpu.json 2>/dev/null
MOSAIC_BENCH_FORCE_DIST=1 python3 bench.py --steps 5 --warmup 2 --no-cpu > $O/bench_D_dist1.json 2> $O/bench_D_dist1.err
python3 bench.py --steps 5 --warmup 2 --no-cpu --opt lds_filter=1 --opt tile=128 > $O/bench_D_lds_tile.json 2>/dev/null
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=fast -fno-slp-vectorize -Iinclude -c mosaicmagnifique_amd/csrc/filter.hip -o /tmp/filter_ru.o -Rpass-analysis=kernel-resource-usage 2> $O/filter_resource_usage.txt
find $O -name "*.csv" | head -30
