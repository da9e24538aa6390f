# Round 4: the host's wait -- polling the stream for up to 1 ms before blocking on it, against blocking at once (same box, alternating).
set -x
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_spin; mkdir -p $O
for rep in 1 2 3; do
  for c in A B H F C P D; do
    for sp in 0 1000; do
      python3 bench.py --config $c --steps 30 --warmup 5 --no-cpu --opt spin_wait_us=$sp > $O/x.json 2>/dev/null; python3 tools/experiments/r4_line.py $O/x.json "$c spin_wait_us=$sp"
    done
  done
done
timeout -k 10 300 python3 -m pytest tests/test_gpu_generator.py tests/test_gpu_first_call.py tests/test_gpu_small_jobs.py -x -q > $O/pytest.log 2>&1; echo "pytest rc $?"; tail -2 $O/pytest.log
