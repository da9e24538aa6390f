# Round 5: the short lists' length once more (spec_k 2 / 3 / 4 (default) / 6, spec_auto off) on configs D and P.
set -x
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_speck; mkdir -p $O
for cfg in D P; do
  for k in 2 3 4 6; do
    python3 bench.py --config $cfg --steps 8 --warmup 3 --no-cpu --opt spec_k=$k --opt spec_auto=0 > $O/${cfg}_k$k.json 2>$O/err.log || { tail -5 $O/err.log; exit 1; }
  done
done
python3 - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r05_speck/*.json')):
    d=json.loads(open(f).read().strip().splitlines()[-1]); t=d.get('last_step_device_ms',{})
    print("%-8s %8.3f ms  score %7.3f rescore %6.3f visited %.4f deepened %s spec_rounds %s digest %s" % (f.split('/')[-1][:-5], d['ms_per_step'], t.get('score_ms',0), t.get('rescore_ms',0), d['roofline'].get('visited_fraction_of_groups') or 0, t.get('deepened_cells'), t.get('spec_rounds'), str(d.get('result_digest'))[:10]))
PY
