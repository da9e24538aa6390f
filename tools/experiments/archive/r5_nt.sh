# Round 5: the library's groups as non-temporal loads (-DMM_LIB_NT=1) against the default build, A/B on one box.
set -x
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_nt; mkdir -p $O
bash tools/build_variant.sh r5_nt -DMM_LIB_NT=1 > $O/build.log 2>&1 || { tail -5 $O/build.log; exit 1; }
for rep in 1 2; do
  for cfg in D G I E8 C P; do
    python3 bench.py --config $cfg --steps 10 --warmup 3 --no-cpu > $O/${cfg}_base_$rep.json 2>$O/err.log || { tail -5 $O/err.log; exit 1; }
    MOSAIC_HIP_LIB=gpurun_out/r5_nt.so python3 bench.py --config $cfg --steps 10 --warmup 3 --no-cpu > $O/${cfg}_nt_$rep.json 2>$O/err.log || { tail -5 $O/err.log; exit 1; }
  done
done
for rep in 1 2; do
  python3 bench.py --config G --steps 10 --warmup 3 --no-cpu --opt xcd_cells=2 --opt tile=16 > $O/Gx2t16_base_$rep.json 2>$O/err.log
  MOSAIC_HIP_LIB=gpurun_out/r5_nt.so python3 bench.py --config G --steps 10 --warmup 3 --no-cpu --opt xcd_cells=2 --opt tile=16 > $O/Gx2t16_nt_$rep.json 2>$O/err.log
done
python3 - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r05_nt/*_*_?.json')):
    d=json.loads(open(f).read().strip().splitlines()[-1]); t=d.get('last_step_device_ms',{})
    print("%-16s %8.3f ms  score %7.3f digest %s" % (f.split('/')[-1][:-5], d['ms_per_step'], t.get('score_ms',0), d['result_digest'][:10]))
PY
