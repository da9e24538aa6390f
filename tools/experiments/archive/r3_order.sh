# Round 3: long items (clipped cells) first in each XCD's share (option cell_order), A/B per configuration in one call.
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03_order; mkdir -p $O
run() { n=$1; c=$2; k=$3; shift 3
  python3 bench.py --config $c --steps $k --warmup 3 --no-cpu "$@" > $O/bench_$n.json 2> $O/bench_$n.err || echo "FAILED $n"
  python3 - $O/bench_$n.json $n <<'PY'
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1]); t = d["last_step_device_ms"]
    print("%-16s %9.0f cells/s %8.4f ms/step filter %.3f stage %.3f seed %.3f rescore %.3f visited %d dropped %d digest %s" % (sys.argv[2], d["value"], d["ms_per_step"], t["score_ms"], t["stage_ms"], t["seed_ms"], t["rescore_ms"], t["visited_groups"], t["bound_dropped"], d.get("result_digest", "")[:12]))
except Exception as e: print(sys.argv[2], "no line:", e)
PY
}
for o in 1 0 1 0; do run C_order_$o C 60 --opt cell_order=$o --opt phase_timing=1; done
for o in 1 0 1 0; do run E8_order_$o E8 6 --opt cell_order=$o; done
for o in 1 0 1 0; do run D_order_$o D 6 --opt cell_order=$o; done
