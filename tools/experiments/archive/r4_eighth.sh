# Round 4, eighth GPU call: the whole suite on the final sources (few deepened cells on the fp64-only path), config P.
set -x
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_eighth; mkdir -p $O
timeout -k 10 1000 python3 -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log; tail -6 $O/pytest.log | cut -c1-300
for rep in 1 2; do python3 bench.py --config P --steps 8 --warmup 2 --no-cpu > $O/P_$rep.json 2>/dev/null; python3 - $O/P_$rep.json <<'PY'
import json, sys
d = json.load(open(sys.argv[1])); t = d["last_step_device_ms"]
print("P ms/step %.3f" % d["ms_per_step"], {k: t[k] for k in ("seed_ms", "score_ms", "rescore_ms", "select_ms", "resolve_ms", "spec_rounds", "deepened_cells", "dense_cells", "host_syncs", "score_launches")})
PY
done
