# Round 3: what the single-submission path and the fused kernels buy, per configuration (A/B pairs in one call on
# one box).  usage (on the box): bash tools/experiments/r3_small_jobs.sh TAG   -> gpurun_out/TAG/
V=${1:-r03_small}
cd $GRAFT_REPO_ROOT
O=gpurun_out/$V
mkdir -p $O
run() { # name config steps opts...
  n=$1; c=$2; k=$3; shift 3
  python3 bench.py --config $c --steps $k --warmup 5 --no-cpu "$@" > $O/bench_$n.json 2> $O/bench_$n.err || echo "FAILED $n"
  python3 - $O/bench_$n.json $n <<'PY'
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
    t = d["last_step_device_ms"]
    print("%-22s %9.0f cells/s  %8.4f ms/step  syncs %d fast %d  stage %.3f seed %.3f score %.3f rescore %.3f select %.3f resolve %.3f total %.3f dense %d tile %d" % (
        sys.argv[2], d["value"], d["ms_per_step"], t["host_syncs"], t["fast_path"], t["stage_ms"], t["seed_ms"], t["score_ms"],
        t["rescore_ms"], t["select_ms"], t["resolve_ms"], t["total_ms"], t["dense_cells"], t["filter_tile"]))
except Exception as e:
    print(sys.argv[2], "no line:", e)
PY
}
for spec in "$@"; do :; done
run A_default A 500
run A_events A 500 --opt phase_timing=1
run A_pruned A 500 --opt prune_min_kpairs=0
run A_slow A 500 --opt fast_path=0 --opt fused=0 --opt fused_seeds=0 --opt prune_min_kpairs=0 --opt phase_timing=1
run B_default B 200
run B_noevents B 200 --opt phase_timing=0
run B_seeds3 B 200 --opt fused_seeds=0
run B_tile128 B 200 --opt tile=128
run B_tile256 B 200 --opt tile=256
run B_tile1000 B 200 --opt tile=1000
run B_slow B 200 --opt fast_path=0 --opt fused=0 --opt fused_seeds=0
run C_default C 100
run C_seeds3 C 100 --opt fused_seeds=0
run C_slow C 100 --opt fast_path=0 --opt fused=0 --opt fused_seeds=0
run D_default D 10
run D_slow D 10 --opt fast_path=0 --opt fused=0
