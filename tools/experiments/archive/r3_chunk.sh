cd $GRAFT_REPO_ROOT
O=gpurun_out/r03_chunk; mkdir -p $O
run() { n=$1; c=$2; k=$3; shift 3
  python3 bench.py --config $c --steps $k --warmup 3 --no-cpu "$@" > $O/bench_$n.json 2> $O/bench_$n.err || echo "FAILED $n"
  python3 - $O/bench_$n.json $n <<'PY'
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1]); t = d["last_step_device_ms"]
    print("%-16s %9.0f cells/s %8.4f ms/step score %.3f seed %.3f rescore %.3f total %.3f visited %d" % (sys.argv[2], d["value"], d["ms_per_step"], t["score_ms"], t["seed_ms"], t["rescore_ms"], t["total_ms"], t["visited_groups"]))
except Exception as e: print(sys.argv[2], "no line:", e)
PY
}
for ch in 1 0 2 4 8 16; do run C_chunk$ch C 60 --opt item_chunk=$ch --opt phase_timing=1; done
for ch in 1 0 4 8; do run E8_chunk$ch E8 6 --opt item_chunk=$ch; done
for ch in 1 0 4; do run D_chunk$ch D 6 --opt item_chunk=$ch; done
python3 bench.py --sweep > $O/sweep.json 2> $O/sweep.err; tail -20 $O/sweep.err
