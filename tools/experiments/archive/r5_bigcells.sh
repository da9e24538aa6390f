# Round 5: configs G and I press on the fabric (0.95 / 0.76 of the 8 TB/s peak without the pre-tier): XCD sub-group size and item size again.
set -x
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_bigcells; mkdir -p $O
for cfg in G I; do
  for xc in 0 2 4 8 16 32; do
    for tile in 0 16 64; do
      python3 bench.py --config $cfg --steps 10 --warmup 3 --no-cpu --opt pretier=0 --opt xcd_cells=$xc --opt tile=$tile > $O/${cfg}_x${xc}_t${tile}.json 2>$O/err.log || { tail -5 $O/err.log; exit 1; }
    done
  done
done
python3 - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r05_bigcells/*.json')):
    d=json.loads(open(f).read().strip().splitlines()[-1]); t=d.get('last_step_device_ms',{})
    print("%-16s %8.3f ms  score %7.3f tile %s" % (f.split('/')[-1][:-5], d['ms_per_step'], t.get('score_ms',0), t.get('filter_tile')))
PY
