# Round 4, first GPU call: the whole -m gpu suite (new: whole-grid goldens, adversarial band maximiser, four masks of
# different areas, plain `bench.py --gpus 2`), then the cost of the dense band guard as A/B pairs on one box, then the
# bench lines with the floor / first-call legs.
set -x
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_first; mkdir -p $O
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log; tail -5 $O/pytest.log
for c in D C; do
  for g in 1 64 1 64; do
    python3 bench.py --config $c --steps 10 --warmup 2 --no-cpu --opt band_guard=$g > $O/guard_${c}_$g.json 2>/dev/null
    python3 - $O/guard_${c}_$g.json $c $g <<'PY'
import json, sys
d = json.load(open(sys.argv[1])); t = d["last_step_device_ms"]
print("guard", sys.argv[2], sys.argv[3], "ms/step %.3f" % d["ms_per_step"], "score %.3f rescore %.3f select %.3f" % (t["score_ms"], t["rescore_ms"], t["select_ms"]), "alarms", t["band_alarms"])
PY
  done
done > $O/guard_ab.txt 2>&1
cat $O/guard_ab.txt
python3 bench.py --steps 10 --warmup 2 > $O/bench_D.json 2> $O/bench_D.err
for c in A B C H G P; do python3 bench.py --config $c --steps 5 --warmup 2 > $O/bench_$c.json 2> $O/bench_$c.err; done
python3 - $O <<'PY'
import json, sys, glob
for p in sorted(glob.glob(sys.argv[1] + "/bench_*.json")):
    try:
        d = json.load(open(p))
    except Exception as e:
        print(p, "unreadable", e); continue
    fc = d.get("first_call", {})
    print(p.split("/")[-1], "cells/s %.0f ms/step %.3f" % (d["value"], d["ms_per_step"]),
          "floors", {k: round(d[k]["ms_per_step"], 2) for k in ("floor_unpruned", "floor_fp64_only") if k in d},
          "first", {k: (round(fc[k]["first_call_ms"], 2), fc[k]["engine_clocks"]) for k in ("first", "again") if k in fc},
          "parity", d.get("parity_on_cpu_sample", {}).get("indices_equal"))
PY
