set -x
V=${1:-v8}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/$V
python3 bench.py --steps 5 --warmup 2 > gpurun_out/$V/bench_D.json 2> gpurun_out/$V/bench_D.err
for c in A B C E8; do python3 bench.py --config $c --steps 5 --warmup 2 --no-cpu > gpurun_out/$V/bench_$c.json 2>/dev/null; done
MOSAIC_BENCH_FORCE_DIST=1 python3 bench.py --steps 5 --warmup 2 --no-cpu > gpurun_out/$V/bench_D_dist1.json 2> gpurun_out/$V/bench_D_dist1.err
rocprofv3 --kernel-trace --stats -d gpurun_out/$V/kt -o d --output-format csv -- python3 bench.py --steps 3 --warmup 1 --no-cpu > gpurun_out/$V/kt.log 2>&1
rocprofv3 --pmc FETCH_SIZE -d gpurun_out/$V/fetch -o p --output-format csv -- python3 bench.py --config D --steps 1 --warmup 0 --no-cpu > gpurun_out/$V/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d gpurun_out/$V/write -o p --output-format csv -- python3 bench.py --config D --steps 1 --warmup 0 --no-cpu > gpurun_out/$V/write.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU GRBM_GUI_ACTIVE -d gpurun_out/$V/sq -o p --output-format csv -- python3 bench.py --config D --steps 1 --warmup 0 --no-cpu > gpurun_out/$V/sq.log 2>&1
find gpurun_out/$V -name "*.csv" | head -20
