# Round 4: the audit for the Euclidean filter -- tests, goldens with every drop audited, cost on configs B and F.
set -x
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_euclid_audit; mkdir -p $O
timeout -k 10 300 python3 -m pytest tests/test_gpu_filter.py -x -q -s -k "audit" > $O/pytest.log 2>&1; echo "pytest rc $?"; grep -a "^broken\|^\.broken\|passed\|failed\|Error" $O/pytest.log | cut -c1-250
timeout -k 10 600 python3 -m pytest tests/test_golden_grids.py -x -q -s -m gpu -k "audited" > $O/pytest2.log 2>&1; echo "pytest2 rc $?"; grep -a "audited\|passed\|failed\|assert" $O/pytest2.log | cut -c1-200
for rep in 1 2 3; do
  for c in B F C H D; do
    for a in 0 1024; do
      python3 bench.py --config $c --steps 20 --warmup 3 --no-cpu --opt audit=$a > $O/x.json 2>/dev/null; python3 tools/experiments/r4_line.py $O/x.json "$c audit=$a"
    done
  done
done
