# Round 4, final pass B: config E's PMC passes, the kernel trace, the bench lines of every configuration, the harness sweep.
set -x
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
CONFIGS="E" bash tools/profile.sh r04_v48
python3 bench.py --sweep > gpurun_out/r04_v48/sweep.json 2> gpurun_out/r04_v48/sweep.err; tail -16 gpurun_out/r04_v48/sweep.err | cut -c1-220
