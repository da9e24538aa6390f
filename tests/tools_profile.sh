set -x
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/v7
python3 bench.py --steps 5 --warmup 2 > gpurun_out/v7/bench_D.json 2> gpurun_out/v7/bench_D.err
for c in A B C E8; do python3 bench.py --config $c --steps 5 --warmup 2 --no-cpu > gpurun_out/v7/bench_$c.json 2>/dev/null; done
rocprofv3 --kernel-trace --stats -d gpurun_out/v7/kt -o d --output-format csv -- python3 bench.py --steps 3 --warmup 1 --no-cpu > gpurun_out/v7/kt.log 2>&1
rocprofv3 --pmc FETCH_SIZE -d gpurun_out/v7/fetch -o p --output-format csv -- python3 bench.py --config D --steps 1 --warmup 0 --no-cpu > gpurun_out/v7/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d gpurun_out/v7/write -o p --output-format csv -- python3 bench.py --config D --steps 1 --warmup 0 --no-cpu > gpurun_out/v7/write.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU GRBM_GUI_ACTIVE -d gpurun_out/v7/sq -o p --output-format csv -- python3 bench.py --config D --steps 1 --warmup 0 --no-cpu > gpurun_out/v7/sq.log 2>&1
find gpurun_out/v7 -name "*.csv" | head -20
