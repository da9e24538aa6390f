# Round 3: how many proxy-chosen seeds config B's limit wants (option "seeds": extra seeds beyond the list length).
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03_bseeds; mkdir -p $O
for sd in 14 10 6 2 0; do
  python3 bench.py --config B --steps 200 --warmup 5 --no-cpu --opt seeds=$sd > $O/bench_B_seeds$sd.json 2>/dev/null
  python3 - $O/bench_B_seeds$sd.json $sd <<'PY'
import json, sys
d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1]); t = d["last_step_device_ms"]
print("B seeds=%-3s %9.0f cells/s %8.4f ms/step  seed %.3f filter %.3f re-score %.3f" % (sys.argv[2], d["value"], d["ms_per_step"], t["seed_ms"], t["score_ms"], t["rescore_ms"]))
PY
done
