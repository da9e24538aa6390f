# Round 4, sixth GPU call: extra seeds on the large CIEDE2000 configurations (config C ran flat from 6 to 14, config H best at 2);
# the bench lines' whole-grid golden parity.
set -x
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_sixth; mkdir -p $O
line() { python3 - "$@" <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1])); t = d["last_step_device_ms"]; r = d["roofline"]
    print(sys.argv[2], "ms/step %.3f" % d["ms_per_step"], "kernel %.3f" % r["avg_launch_ms"], "visited %.4f" % (r["visited_fraction_of_groups"] or 0),
          "seed %.3f rescore %.3f" % (t["seed_ms"], t["rescore_ms"]), d.get("parity_on_golden"), (d.get("parity_on_cpu_sample") or {}).get("sample", "")[:90])
except Exception as e:
    print(sys.argv[2], "unreadable", e)
PY
}
for rep in 1 2; do
for c in D E8 G I; do for sd in 6 10 14; do python3 bench.py --config $c --steps 8 --warmup 2 --no-cpu --opt seeds=$sd > $O/${c}_seeds${sd}_$rep.json 2>/dev/null; line $O/${c}_seeds${sd}_$rep.json "$c seeds=$sd"; done; done
done > $O/seeds.txt 2>&1
cat $O/seeds.txt
for c in D C B; do python3 bench.py --config $c --steps 5 --warmup 2 > $O/bench_$c.json 2> $O/bench_$c.err; line $O/bench_$c.json "bench $c"; done
