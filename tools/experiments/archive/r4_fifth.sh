# Round 4, fifth GPU call: the whole suite on the tree with the pre-tier by cell size, then the bench lines.
set -x
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_fifth; mkdir -p $O
timeout -k 10 1000 python3 -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log; tail -6 $O/pytest.log | cut -c1-300
line() { python3 - "$@" <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1])); t = d["last_step_device_ms"]; r = d["roofline"]
    fc = d.get("first_call", {})
    print(sys.argv[2], "cells/s %.0f ms/step %.3f" % (d["value"], d["ms_per_step"]), "kernel %.3f" % r["avg_launch_ms"], "visited %.4f" % (r["visited_fraction_of_groups"] or 0),
          "seed %.3f rescore %.3f resolve %.3f rounds %d deepened %d syncs %d" % (t["seed_ms"], t["rescore_ms"], t["resolve_ms"], t["spec_rounds"], t["deepened_cells"], t["host_syncs"]),
          "floors", {k: round(d[k]["ms_per_step"], 2) for k in ("floor_unpruned", "floor_fp64_only") if k in d},
          "first/again", [round(fc[k]["first_call_ms"], 2) for k in ("first", "again") if k in fc], "parity", d.get("parity_on_cpu_sample", {}).get("indices_equal"))
except Exception as e:
    print(sys.argv[2], "unreadable", e)
PY
}
python3 bench.py --steps 10 --warmup 2 > $O/bench_D.json 2> $O/bench_D.err; line $O/bench_D.json "bench D"
for c in A B C E8 F G H I P; do python3 bench.py --config $c --steps 5 --warmup 2 > $O/bench_$c.json 2> $O/bench_$c.err; line $O/bench_$c.json "bench $c"; done
