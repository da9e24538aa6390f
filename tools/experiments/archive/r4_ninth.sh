# Round 4, ninth GPU call: the whole-grid goldens of every bench configuration (A, F, G, H, I, P added), then their bench lines
# (parity_on_golden in each).
set -x
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_ninth; mkdir -p $O
timeout -k 10 900 python3 -m pytest tests/test_golden_grids.py -m gpu -x -q -s > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log; grep -v "^$" $O/pytest.log | tail -16 | cut -c1-300
for c in A F G H I P; do python3 bench.py --config $c --steps 5 --warmup 2 > $O/bench_$c.json 2> $O/bench_$c.err; python3 - $O/bench_$c.json $c <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print("bench", sys.argv[2], "ms/step %.3f" % d["ms_per_step"], d.get("parity_on_golden"), d["parity_on_cpu_sample"]["indices_equal"])
PY
done
