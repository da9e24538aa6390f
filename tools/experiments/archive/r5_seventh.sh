# Round 5, seventh pass: the suite on the big-cell defaults; G / I / D / F step times.
set -x
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_seventh; mkdir -p $O
timeout -k 10 1000 python3 -m pytest tests -m gpu -x -q -p no:cacheprovider > $O/suite.log 2>&1; rc=$?; echo "suite rc $rc"; tail -3 $O/suite.log
[ $rc -eq 0 ] || { grep -E "^(FAILED|E  )" $O/suite.log | head -20; exit $rc; }
for cfg in G I D; do
  for opts in "" "--opt pretier=1" "--opt pretier=0"; do
    tag=$(echo "$opts" | tr -d ' -' | tr '=' '_'); tag=${tag:-auto}
    python3 bench.py --config $cfg --steps 10 --warmup 4 --no-cpu $opts > $O/${cfg}_$tag.json 2>$O/err.log || { tail -5 $O/err.log; exit 1; }
  done
done
python3 - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r05_seventh/*_*.json')):
    d=json.loads(open(f).read().strip().splitlines()[-1]); t=d.get('last_step_device_ms',{})
    print("%-22s %8.3f ms  score %7.3f tile %s pretier %s" % (f.split('/')[-1][:-5], d['ms_per_step'], t.get('score_ms',0), t.get('filter_tile'), t.get('pretier')))
PY
