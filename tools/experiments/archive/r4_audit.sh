# Round 4: the audit of every way a candidate leaves the CIEDE2000 filter without a full score -- its test, the goldens, and what
# it costs per step at 1 in 1024 / 256 / off (same box, alternating).
set -x
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_audit; mkdir -p $O
timeout -k 10 300 python3 -m pytest tests/test_gpu_filter.py -x -q -s -k "audit" > $O/pytest.log 2>&1; echo "pytest rc $?"; grep -a "broken\|passed\|failed\|Error\|assert" $O/pytest.log | cut -c1-250
timeout -k 10 600 python3 -m pytest tests/test_golden_grids.py tests/test_gpu_fullsize.py -x -q -m gpu > $O/pytest2.log 2>&1; echo "pytest2 rc $?"; tail -3 $O/pytest2.log
for rep in 1 2; do
  for c in D E8 G C; do
    for a in 0 1024 256; do
      python3 bench.py --config $c --steps 8 --warmup 2 --no-cpu --opt audit=$a > $O/x.json 2>/dev/null; python3 tools/experiments/r4_line.py $O/x.json "$c audit=$a"
    done
  done
done
