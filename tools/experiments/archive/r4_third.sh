# Round 4, third GPU call: suite; the big-cell form as its own template instantiation (config D must equal the round-3 form:
# A/B against -DMM_BIG_CELL_STAGE0=0 in one call); config P after the single-submission resolve of later rounds, and with longer
# speculative lists; extra seeds on the small CIEDE2000 jobs; the packed-fp16 form's arithmetic rate; the
# adversarial maximiser's and the structured sweep's printed results.
set -x
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_third; mkdir -p $O
timeout -k 10 1000 python3 -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log; tail -5 $O/pytest.log
timeout -k 10 300 python3 -m pytest tests/test_gpu_filter.py -m gpu -q -s -k "adversarial or structured" > $O/band_maximiser.txt 2>&1
grep -i "adversarial\|band sweep" $O/band_maximiser.txt
bash tools/build_variant.sh big0 -DMM_BIG_CELL_STAGE0=0 > $O/build_big0.log 2>&1
line() { python3 - "$@" <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1])); t = d["last_step_device_ms"]; r = d["roofline"]
    print(sys.argv[2], "ms/step %.3f" % d["ms_per_step"], "kernel %.3f" % r["avg_launch_ms"], "visited %.4f" % (r["visited_fraction_of_groups"] or 0),
          "seed %.3f rescore %.3f resolve %.3f rounds %d deepened %d syncs %d" % (t["seed_ms"], t["rescore_ms"], t["resolve_ms"], t["spec_rounds"], t["deepened_cells"], t["host_syncs"]))
except Exception as e:
    print(sys.argv[2], "unreadable", e)
PY
}
for rep in 1 2 3; do
  for c in D; do
    python3 bench.py --config $c --steps 10 --warmup 2 --no-cpu > $O/ab_${c}_new_$rep.json 2>/dev/null; line $O/ab_${c}_new_$rep.json "$c template"
    MOSAIC_HIP_LIB=gpurun_out/big0.so python3 bench.py --config $c --steps 10 --warmup 2 --no-cpu > $O/ab_${c}_old_$rep.json 2>/dev/null; line $O/ab_${c}_old_$rep.json "$c macro=0"
  done
done > $O/big_ab.txt 2>&1
for c in G I; do python3 bench.py --config $c --steps 8 --warmup 2 --no-cpu > $O/ab_$c.json 2>/dev/null; line $O/ab_$c.json "$c template"; done >> $O/big_ab.txt 2>&1
cat $O/big_ab.txt
for k in 4 6 8 12; do python3 bench.py --config P --steps 8 --warmup 2 --no-cpu --opt spec_k=$k > $O/P_spec$k.json 2>/dev/null; line $O/P_spec$k.json "P spec_k=$k"; done > $O/P_spec.txt 2>&1
cat $O/P_spec.txt
for c in C H; do for sd in 2 6 10 14 22; do python3 bench.py --config $c --steps 10 --warmup 3 --no-cpu --opt seeds=$sd > $O/${c}_seeds$sd.json 2>/dev/null; line $O/${c}_seeds$sd.json "$c seeds=$sd"; done; done > $O/seeds.txt 2>&1
cat $O/seeds.txt
hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=fast -fno-slp-vectorize -Imosaicmagnifique_amd/csrc -o /tmp/fp16_form tools/experiments/r4_fp16_form.hip > $O/fp16_build.log 2>&1 && timeout -k 10 120 /tmp/fp16_form > $O/fp16_form.txt 2>&1
cat $O/fp16_form.txt
python3 bench.py --config P --steps 8 --warmup 2 > $O/bench_P.json 2> $O/bench_P.err; line $O/bench_P.json "bench P"
