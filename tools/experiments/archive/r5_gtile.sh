# Round 5: config G's item size between 16 and 32 candidates (a quarter of its wave cycles wait at the item's last barrier), with the XCD sub-group at 2 / 4 cells.
set -x
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_gtile; mkdir -p $O
for xc in 0 4; do
  for tile in 12 16 20 24 28 32; do
    python3 bench.py --config G --steps 10 --warmup 3 --no-cpu --opt xcd_cells=$xc --opt tile=$tile > $O/G_x${xc}_t${tile}.json 2>$O/err.log || { tail -5 $O/err.log; exit 1; }
  done
done
python3 - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r05_gtile/*.json')):
    d=json.loads(open(f).read().strip().splitlines()[-1]); t=d.get('last_step_device_ms',{})
    print("%-16s %8.3f ms  score %7.3f tile %s" % (f.split('/')[-1][:-5], d['ms_per_step'], t.get('score_ms',0), t.get('filter_tile')))
PY
