# Round 5, final pass B: config E's PMC passes, the kernel trace, the bench lines of every configuration, the harness sweep.
set -x
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05_v52; CONFIGS="E" bash tools/profile.sh r05_v52 > gpurun_out/r05_v52/profile_b.log 2>&1; tail -3 gpurun_out/r05_v52/profile_b.log
python3 bench.py --sweep > gpurun_out/r05_v52/sweep.json 2> gpurun_out/r05_v52/sweep.err; tail -16 gpurun_out/r05_v52/sweep.err | cut -c1-220
