# Round 5, fourth pass: the algebraic fp64 form in every parallel-order sum: its own test, the whole suite, the step times against r5_third's.
set -x
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_fourth; mkdir -p $O
timeout -k 10 300 python3 -m pytest tests/test_gpu_colour_diff.py -x -q -s -p no:cacheprovider > $O/colour.log 2>&1; rc=$?; echo "colour rc $rc"; tail -4 $O/colour.log | cut -c1-600
[ $rc -eq 0 ] || exit $rc
timeout -k 10 1000 python3 -m pytest tests -m gpu -x -q -p no:cacheprovider > $O/suite.log 2>&1; rc=$?; echo "suite rc $rc"; tail -3 $O/suite.log
[ $rc -eq 0 ] || { grep -E "^(FAILED|E  )" $O/suite.log | head -20; exit $rc; }
for cfg in D G I C P H; do
  python3 bench.py --config $cfg --steps 12 --warmup 4 --no-cpu > $O/${cfg}.json 2>$O/err.log || { tail -5 $O/err.log; exit 1; }
done
python3 - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r05_fourth/?.json')):
    d=json.loads(open(f).read().strip().splitlines()[-1]); t=d.get('last_step_device_ms',{})
    print("%-8s %8.3f ms  score %7.3f rescore %6.3f seed %6.3f pretier %s digest %s" % (f.split('/')[-1], d['ms_per_step'], t.get('score_ms',0), t.get('rescore_ms',0), t.get('seed_ms',0), t.get('pretier'), d['result_digest'][:12]))
PY
