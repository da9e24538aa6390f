# Round 5: big cells (above 16 groups) walk an item's stage-0 survivors by ascending bound (default) against candidate order (-DMM_BIG_LONG_FIRST=0).
set -x
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
bash tools/build_variant.sh lf0 -DMM_BIG_LONG_FIRST=0 > /dev/null 2>&1 || exit 1
O=gpurun_out/r05_longfirst; mkdir -p $O
for rep in 1 2; do
for cfg in ${CONFIGS:-G I F}; do
  MOSAIC_HIP_LIB=$PWD/gpurun_out/lf0.so python3 bench.py --config $cfg --steps 10 --warmup 3 --no-cpu > $O/${cfg}_lf0_$rep.json 2>$O/err.log || { tail -5 $O/err.log; exit 1; }
  python3 bench.py --config $cfg --steps 10 --warmup 3 --no-cpu > $O/${cfg}_lf1_$rep.json 2>$O/err.log || { tail -5 $O/err.log; exit 1; }
done
done
python3 - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r05_longfirst/*.json')):
    d=json.loads(open(f).read().strip().splitlines()[-1]); t=d.get('last_step_device_ms',{})
    print("%-12s %8.3f ms  score %7.3f seed %6.3f rescore %6.3f visited %.4f digest %s" % (f.split('/')[-1][:-5], d['ms_per_step'], t.get('score_ms',0), t.get('seed_ms',0), t.get('rescore_ms',0), d['roofline'].get('visited_fraction_of_groups') or 0, str(d.get('result_digest'))[:12]))
PY
