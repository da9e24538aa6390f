# Round 5: the limit follows the full scores down during the launch ("tighten" 1) against the seeds-only limit (0): configs D, E8, G, P, C, I,
# interleaved A/B/A/B; then the parity tests of the filter.
set -x
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_tighten; mkdir -p $O
for rep in 1 2; do
for cfg in D E8 G P C I; do
  for t in 0 1; do
    python3 bench.py --config $cfg --steps 10 --warmup 3 --no-cpu --opt tighten=$t > $O/${cfg}_t${t}_$rep.json 2>$O/err.log || { tail -5 $O/err.log; exit 1; }
  done
done
done
python3 - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r05_tighten/*.json')):
    d=json.loads(open(f).read().strip().splitlines()[-1]); t=d.get('last_step_device_ms',{})
    print("%-12s %8.3f ms  score %7.3f seed %6.3f rescore %6.3f visited %.4f parity %s" % (f.split('/')[-1][:-5], d['ms_per_step'], t.get('score_ms',0), t.get('seed_ms',0), t.get('rescore_ms',0), d['roofline'].get('visited_fraction_of_groups') or 0, d.get('parity')))
PY
timeout -k 10 900 python3 -m pytest tests/test_gpu_filter.py tests/test_gpu_generator.py tests/test_gpu_fuzz.py -m gpu -x -q 2>&1 | tail -5
