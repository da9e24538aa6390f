# Round 5, third pass: the whole GPU suite on the worst-case band build (pre-tier trial with a warm-up launch, eligibility by cell size), then the step times of the configs.
set -x
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_third; mkdir -p $O
timeout -k 10 1000 python3 -m pytest tests -m gpu -x -q -p no:cacheprovider > $O/suite.log 2>&1; rc=$?; echo "suite rc $rc"; tail -3 $O/suite.log
[ $rc -eq 0 ] || exit $rc
for cfg in D G I C P H; do
  python3 bench.py --config $cfg --steps 12 --warmup 4 --no-cpu > $O/${cfg}.json 2>$O/err.log || { tail -5 $O/err.log; exit 1; }
done
python3 - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r05_third/?.json')):
    d=json.loads(open(f).read().strip().splitlines()[-1]); t=d.get('last_step_device_ms',{})
    print("%-8s %8.3f ms  score %7.3f rescore %6.3f seed %6.3f pretier %s" % (f.split('/')[-1], d['ms_per_step'], t.get('score_ms',0), t.get('rescore_ms',0), t.get('seed_ms',0), t.get('pretier')))
PY
