# Round 4: the fp16 form after the maximiser's first find (C' floored by the stored chroma) and with one transcendental per
# 7th-power ratio: the band sweeps and the maximiser, the goldens of the configurations that run the pre-tier, what it costs per step.
set -x
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_ratio; mkdir -p $O
timeout -k 10 600 python3 -m pytest tests/test_gpu_filter.py -x -q -s -k "pretier_band or maximiser" > $O/pytest.log 2>&1; echo "pytest rc $?"; grep -a "sweep\|maximiser\|passed\|failed" $O/pytest.log | cut -c1-250
timeout -k 10 600 python3 -m pytest tests/test_golden_grids.py tests/test_gpu_fullsize.py -x -q -m gpu > $O/pytest2.log 2>&1; echo "pytest2 rc $?"; tail -3 $O/pytest2.log
for rep in 1 2; do
  for c in D E8 G I; do
      python3 bench.py --config $c --steps 8 --warmup 2 --no-cpu --opt pretier=1 > $O/x.json 2>/dev/null; python3 tools/experiments/r4_line.py $O/x.json "$c pretier=1"
  done
done
