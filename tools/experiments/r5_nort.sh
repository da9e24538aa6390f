# Round 5: pixel groups with a provably negligible rotation term evaluated without it (-DMM_NORT=1), A/B on one box; goldens on the variant.
set -x
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_nort; mkdir -p $O
bash tools/build_variant.sh r5_nort -DMM_NORT=1 > $O/build.log 2>&1 || { tail -5 $O/build.log; exit 1; }
MOSAIC_HIP_LIB=gpurun_out/r5_nort.so timeout -k 10 600 python3 -m pytest tests/test_golden_grids.py tests/test_gpu_filter.py -x -q -p no:cacheprovider > $O/tests.log 2>&1; rc=$?; echo "tests rc $rc"; tail -3 $O/tests.log | cut -c1-200
[ $rc -eq 0 ] || exit $rc
for rep in 1 2; do
  for cfg in D E8 C P; do
    python3 bench.py --config $cfg --steps 10 --warmup 3 --no-cpu > $O/${cfg}_base_$rep.json 2>$O/err.log || { tail -5 $O/err.log; exit 1; }
    MOSAIC_HIP_LIB=gpurun_out/r5_nort.so python3 bench.py --config $cfg --steps 10 --warmup 3 --no-cpu > $O/${cfg}_nort_$rep.json 2>$O/err.log || { tail -5 $O/err.log; exit 1; }
  done
done
python3 - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r05_nort/*_*_?.json')):
    d=json.loads(open(f).read().strip().splitlines()[-1]); t=d.get('last_step_device_ms',{})
    print("%-16s %8.3f ms  score %7.3f digest %s" % (f.split('/')[-1][:-5], d['ms_per_step'], t.get('score_ms',0), d['result_digest'][:10]))
PY
