# Round 5, fifth pass: the trimmed algebraic form (one division, reciprocal roots for the 7th-power ratios, a polynomial for sin(2 d_theta)).
set -x
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_fifth; mkdir -p $O
timeout -k 10 600 python3 -m pytest tests/test_gpu_colour_diff.py tests/test_golden_grids.py tests/test_gpu_generator.py -x -q -s -p no:cacheprovider > $O/tests.log 2>&1; rc=$?; echo "tests rc $rc"; tail -3 $O/tests.log | cut -c1-300; grep "algebraic fp64" $O/tests.log | cut -c1-700
[ $rc -eq 0 ] || exit $rc
for cfg in D G I C P H; do
  python3 bench.py --config $cfg --steps 12 --warmup 4 --no-cpu > $O/${cfg}.json 2>$O/err.log || { tail -5 $O/err.log; exit 1; }
done
python3 - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r05_fifth/?.json')):
    d=json.loads(open(f).read().strip().splitlines()[-1]); t=d.get('last_step_device_ms',{})
    print("%-8s %8.3f ms  score %7.3f rescore %6.3f seed %6.3f pretier %s digest %s" % (f.split('/')[-1], d['ms_per_step'], t.get('score_ms',0), t.get('rescore_ms',0), t.get('seed_ms',0), t.get('pretier'), d['result_digest'][:12]))
PY
