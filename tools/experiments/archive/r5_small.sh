# Round 5: the small CIEDE2000 jobs (configs H and C) against the filter's item knobs once more.
set -x
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_small; mkdir -p $O
for cfg in H C; do
  for opts in "" "--opt item_chunk=2" "--opt item_chunk=4" "--opt tile=32" "--opt tile=128" "--opt xcd_aware=0" "--opt seeds=0" "--opt seeds=6" "--opt audit=0" "--opt band_guard=0"; do
    tag=$(echo "$opts" | tr -d ' -' | tr '=' '_'); tag=${tag:-default}
    python3 bench.py --config $cfg --steps 30 --warmup 5 --no-cpu $opts > $O/${cfg}_$tag.json 2>$O/err.log || { tail -5 $O/err.log; exit 1; }
  done
done
python3 - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r05_small/*.json')):
    d=json.loads(open(f).read().strip().splitlines()[-1]); t=d.get('last_step_device_ms',{})
    print("%-26s %8.4f ms  score %7.4f seed %6.4f rescore %6.4f" % (f.split('/')[-1][:-5], d['ms_per_step'], t.get('score_ms',0), t.get('seed_ms',0), t.get('rescore_ms',0)))
PY
