# Round 5, second pass: the worst-case band of the fp16 pre-tier + the tiny-chroma rule: filter tests, goldens; what the band costs (A/B by options).
set -x
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_second; mkdir -p $O
timeout -k 10 900 python3 -m pytest tests/test_gpu_filter.py tests/test_golden_grids.py -x -q -s -p no:cacheprovider > $O/tests.log 2>&1; rc=$?; echo "tests rc $rc"; tail -3 $O/tests.log
grep -E "sweep|maximiser" $O/tests.log | cut -c1-260
[ $rc -eq 0 ] || exit $rc
for cfg in D G E I; do
  for opts in "" "--opt pretier=0" "--opt pretier=1" "--opt pretier=1 --opt pretier_band=1"; do
    tag=$(echo "$opts" | tr -d ' -' | tr '=' '_'); tag=${tag:-auto}
    python3 bench.py --config $cfg --steps 12 --warmup 4 --no-cpu $opts > $O/${cfg}_$tag.json 2>$O/err.log || { tail -5 $O/err.log; exit 1; }
  done
done
python3 - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r05_second/*_*.json')):
    d=json.loads(open(f).read().strip().splitlines()[-1]); t=d.get('last_step_device_ms',{})
    print("%-40s %8.3f ms  score %7.3f  pretier %s  visited %.4f" % (f.split('/')[-1], d['ms_per_step'], t.get('score_ms',0), t.get('pretier'), d['roofline'].get('visited_fraction_of_groups') or 0))
PY
