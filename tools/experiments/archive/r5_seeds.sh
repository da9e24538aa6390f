# Round 5: how much the limit's quality is worth: extra seeds 6 / 14 (default) / 30 / 60 / 120 on configs D, E8 and G.
set -x
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_seeds; mkdir -p $O
for cfg in D E8 G; do
  for sd in 6 14 30 60 120; do
    python3 bench.py --config $cfg --steps 8 --warmup 3 --no-cpu --opt seeds=$sd > $O/${cfg}_s$sd.json 2>$O/err.log || { tail -5 $O/err.log; exit 1; }
  done
done
python3 - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r05_seeds/*.json')):
    d=json.loads(open(f).read().strip().splitlines()[-1]); t=d.get('last_step_device_ms',{})
    print("%-12s %8.3f ms  score %7.3f seed %6.3f rescore %6.3f visited %.4f dropped %d" % (f.split('/')[-1][:-5], d['ms_per_step'], t.get('score_ms',0), t.get('seed_ms',0), t.get('rescore_ms',0), d['roofline'].get('visited_fraction_of_groups') or 0, d['roofline'].get('candidates_dropped_by_block_bound') or 0))
PY
