# Round 3: the whole-image bound ahead of the block bound (filter_kernel stage 0), A/B per configuration in one call.
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03_stage0; mkdir -p $O
run() { n=$1; c=$2; k=$3; shift 3
  python3 bench.py --config $c --steps $k --warmup 3 --no-cpu "$@" > $O/bench_$n.json 2> $O/bench_$n.err || echo "FAILED $n"
  python3 - $O/bench_$n.json $n <<'PY'
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1]); t = d["last_step_device_ms"]
    print("%-16s %9.0f cells/s %8.4f ms/step filter %.3f stage %.3f seed %.3f rescore %.3f visited %d dropped %d" % (sys.argv[2], d["value"], d["ms_per_step"], t["score_ms"], t["stage_ms"], t["seed_ms"], t["rescore_ms"], t["visited_groups"], t["bound_dropped"]))
except Exception as e: print(sys.argv[2], "no line:", e)
PY
}
for s0 in 1 0 1 0; do run D_stage0_$s0 D 6 --opt stage0=$s0; done
for s0 in 1 0 1 0; do run E8_stage0_$s0 E8 6 --opt stage0=$s0; done
for s0 in 1 0 1 0; do run C_stage0_$s0 C 60 --opt stage0=$s0 --opt phase_timing=1; done
python3 tools/smooth_bench.py 2>&1 | grep -v amdgpu.ids
