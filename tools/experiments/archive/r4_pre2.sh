# Round 4: the pre-tier, second pass -- band on the sweeps' frontier (0.11 + 6e-3 v, charged as the walk proceeds), the halves of a
# transcendental joined by the SDWA destination select instead of v_pack, one reciprocal for three terms, the next group's
# operands fetched ahead.  Band sweeps and parity first, then A/B pairs (pretier 1 / 0; prefetch on / off) in one call.
set -x
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_pre2; mkdir -p $O
timeout -k 10 300 python3 -m pytest tests/test_gpu_filter.py -m gpu -q -s -x -k "fp16" > $O/pre_band.txt 2>&1; grep -i "pre-tier sweep\|passed\|failed\|Error" $O/pre_band.txt | cut -c1-400
timeout -k 10 900 python3 -m pytest tests/test_golden_grids.py tests/test_gpu_filter.py tests/test_gpu_fullsize.py tests/test_gpu_fuzz.py -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log; tail -5 $O/pytest.log | cut -c1-300
bash tools/build_variant.sh nopf -DMM_PRE_PREFETCH=0 > $O/build_nopf.log 2>&1
line() { python3 - "$@" <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1])); t = d["last_step_device_ms"]; r = d["roofline"]
    print(sys.argv[2], "ms/step %.3f" % d["ms_per_step"], "kernel %.3f" % r["avg_launch_ms"], "visited %.4f" % (r["visited_fraction_of_groups"] or 0), d["result_digest"][:12])
except Exception as e:
    print(sys.argv[2], "unreadable", e)
PY
}
for rep in 1 2; do
  for c in D C E8 G I P; do
    python3 bench.py --config $c --steps 8 --warmup 2 --no-cpu --opt pretier=0 > $O/ab_${c}_0_$rep.json 2>/dev/null; line $O/ab_${c}_0_$rep.json "$c pretier=0"
    python3 bench.py --config $c --steps 8 --warmup 2 --no-cpu --opt pretier=1 > $O/ab_${c}_1_$rep.json 2>/dev/null; line $O/ab_${c}_1_$rep.json "$c pretier=1 prefetch"
    MOSAIC_HIP_LIB=gpurun_out/nopf.so python3 bench.py --config $c --steps 8 --warmup 2 --no-cpu --opt pretier=1 > $O/ab_${c}_1n_$rep.json 2>/dev/null; line $O/ab_${c}_1n_$rep.json "$c pretier=1 no-prefetch"
  done
done > $O/pre_ab.txt 2>&1
cat $O/pre_ab.txt
