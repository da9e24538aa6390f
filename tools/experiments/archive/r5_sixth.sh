# Round 5, sixth pass: sweeps in sets sized by what the previous generate needed, with a device-side early-out; tests of the resolve paths; config P / D.
set -x
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_sixth; mkdir -p $O
timeout -k 10 900 python3 -m pytest tests/test_gpu_generator.py tests/test_golden_grids.py tests/test_gpu_small_jobs.py tests/test_gpu_fuzz.py tests/test_gpu_group.py tests/test_gpu_colour_diff.py -x -q -p no:cacheprovider > $O/tests.log 2>&1; rc=$?; echo "tests rc $rc"; tail -3 $O/tests.log | cut -c1-300
[ $rc -eq 0 ] || { grep -E "^(FAILED|E  )" $O/tests.log | head -20; exit $rc; }
for cfg in P D C; do
  python3 bench.py --config $cfg --steps 12 --warmup 4 --no-cpu > $O/${cfg}.json 2>$O/err.log || { tail -5 $O/err.log; exit 1; }
done
python3 - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r05_sixth/?.json')):
    d=json.loads(open(f).read().strip().splitlines()[-1]); t=d.get('last_step_device_ms',{})
    print("%-8s %8.3f ms  score %7.3f rescore %6.3f resolve %6.3f sweeps %s syncs %s rounds %s digest %s" % (f.split('/')[-1], d['ms_per_step'], t.get('score_ms',0), t.get('rescore_ms',0), t.get('resolve_ms',0), t.get('resolve_sweeps'), t.get('host_syncs'), t.get('spec_rounds'), d['result_digest'][:12]))
PY
