# Round 4: the packed-fp16 pre-tier.  Its band's sweeps, the parity tests with it on, then A/B pairs
# (option pretier 1 / 0) in one call on one box.
set -x
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_pre; mkdir -p $O
timeout -k 10 300 python3 -m pytest tests/test_gpu_filter.py -m gpu -q -s -x -k "fp16" > $O/pre_band.txt 2>&1; grep -i "pre-tier sweep\|passed\|failed\|Error" $O/pre_band.txt | cut -c1-400
timeout -k 10 900 python3 -m pytest tests/test_golden_grids.py tests/test_gpu_filter.py tests/test_gpu_fullsize.py tests/test_gpu_fuzz.py tests/test_gpu_generator.py tests/test_gpu_small_jobs.py tests/test_gpu_first_call.py -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log; tail -12 $O/pytest.log | cut -c1-300
line() { python3 - "$@" <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1])); t = d["last_step_device_ms"]; r = d["roofline"]
    print(sys.argv[2], "ms/step %.3f" % d["ms_per_step"], "kernel %.3f" % r["avg_launch_ms"], "visited %.4f" % (r["visited_fraction_of_groups"] or 0),
          "seed %.3f rescore %.3f resolve %.3f rounds %d deepened %d syncs %d sweeps %d" % (t["seed_ms"], t["rescore_ms"], t["resolve_ms"], t["spec_rounds"], t["deepened_cells"], t["host_syncs"], t["resolve_sweeps"]), d["result_digest"][:12])
except Exception as e:
    print(sys.argv[2], "unreadable", e)
PY
}
for rep in 1 2; do
  for c in D C E8 G I H P; do
    for pre in 1 0; do
      python3 bench.py --config $c --steps 8 --warmup 2 --no-cpu --opt pretier=$pre > $O/ab_${c}_${pre}_$rep.json 2>/dev/null; line $O/ab_${c}_${pre}_$rep.json "$c pretier=$pre"
    done
  done
done > $O/pre_ab.txt 2>&1
cat $O/pre_ab.txt
