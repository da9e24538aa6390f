# Round 5, first pass: the whole GPU suite on the advisor fixes; config D / G A/B of the SDWA block's leading wait state.
set -x
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_first; mkdir -p $O
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q -p no:cacheprovider > $O/suite.log 2>&1; rc=$?; echo "suite rc $rc"; tail -3 $O/suite.log
[ $rc -eq 0 ] || exit $rc
bash tools/build_variant.sh r5_nolead '-DMM_SDWA_LEAD=""' > $O/build.log 2>&1 || { tail -5 $O/build.log; exit 1; }
for rep in 1 2; do
  for cfg in D G; do
    python3 bench.py --config $cfg --steps 20 --warmup 5 --no-cpu > $O/${cfg}_lead_$rep.json 2>$O/err.log || { tail -5 $O/err.log; exit 1; }
    MOSAIC_HIP_LIB=gpurun_out/r5_nolead.so python3 bench.py --config $cfg --steps 20 --warmup 5 --no-cpu > $O/${cfg}_nolead_$rep.json 2>$O/err.log || { tail -5 $O/err.log; exit 1; }
  done
done
python3 - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r05_first/*_*lead_*.json')):
    d=json.loads(open(f).read().strip().splitlines()[-1])
    print(f.split('/')[-1], round(d['ms_per_step'],3), d['dtype'], d.get('last_step_device_ms',{}).get('score_ms'))
PY
