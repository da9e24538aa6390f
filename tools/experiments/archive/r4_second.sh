# Round 4, second GPU call: the suite (whole-grid golden of config D now present; first-call and out-of-box tests), then
# A/B in one call on one box: stage 0 + bound-ordered visits for cells above 16 groups (MM_BIG_CELL_STAGE0 1 vs 0) on
# configs G (128 px), I (80 px) and D (must not move), then config P after the non-fused re-score of deep lists.
set -x
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_second; mkdir -p $O
timeout -k 10 1000 python3 -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log; tail -5 $O/pytest.log
bash tools/build_variant.sh big0 -DMM_BIG_CELL_STAGE0=0 > $O/build_big0.log 2>&1
line() { python3 - "$@" <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1])); t = d["last_step_device_ms"]; r = d["roofline"]
    print(sys.argv[2], "ms/step %.3f" % d["ms_per_step"], "kernel %.3f" % r["avg_launch_ms"], "visited %.4f" % (r["visited_fraction_of_groups"] or 0),
          "seed %.3f rescore %.3f resolve %.3f rounds %d deepened %d" % (t["seed_ms"], t["rescore_ms"], t["resolve_ms"], t["spec_rounds"], t["deepened_cells"]))
except Exception as e:
    print(sys.argv[2], "unreadable", e)
PY
}
for rep in 1 2; do
  for c in G I D; do
    python3 bench.py --config $c --steps 8 --warmup 2 --no-cpu > $O/ab_${c}_new_$rep.json 2>/dev/null; line $O/ab_${c}_new_$rep.json "$c stage0-big=1"
    MOSAIC_HIP_LIB=gpurun_out/big0.so python3 bench.py --config $c --steps 8 --warmup 2 --no-cpu > $O/ab_${c}_old_$rep.json 2>/dev/null; line $O/ab_${c}_old_$rep.json "$c stage0-big=0"
  done
done > $O/big_ab.txt 2>&1
cat $O/big_ab.txt
for c in P C H G I; do python3 bench.py --config $c --steps 5 --warmup 2 > $O/bench_$c.json 2> $O/bench_$c.err; line $O/bench_$c.json "bench $c"; done
