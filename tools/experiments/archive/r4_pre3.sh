# Round 4: the pre-tier's band for the final form -- the frontier of (abs, rel) for the one-reciprocal and the two-reciprocal
# variants, and config D with each.
set -x
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_pre3; mkdir -p $O
cat > /tmp/explore.py <<'PY'
import sys
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from mosaicmagnifique_amd.engine import Engine
e = Engine(0)
grid = 512 * 512 * 24 * 24 * 25
for rel6, abs4 in ((6000, 1100), (6000, 1500), (6000, 2000), (7000, 1300), (7000, 1600), (8000, 1000), (8000, 1300), (8000, 1600), (5000, 2000), (5000, 2500), (4000, 3000)):
    e.set_option("dbg_pre_abs_1e4", abs4); e.set_option("dbg_pre_rel_1e6", rel6)
    worst = 0.0
    for seed in (0, 1, 2):
        worst = max(worst, e.filter_band_sweep(2, grid, seed=seed)["band_share"])
    r3 = e.filter_band_sweep(3, 512 * 128 * 24 * 24 * 25, seed=7)
    print("abs %.3f rel %.4f  cost/px at mean dE 8.5: %.3f  share mode2 %.3f mode3 %.3f" % (abs4 * 1e-4, rel6 * 1e-6, abs4 * 1e-4 + 8.5 * rel6 * 1e-6, worst, r3["band_share"]), flush=True)
PY
bash tools/build_variant.sh tworcp -DMM_PRE_ONE_RCP=0 > $O/build.log 2>&1
echo "one reciprocal" > $O/frontier.txt; python3 /tmp/explore.py >> $O/frontier.txt 2>&1
echo "two reciprocals" >> $O/frontier.txt; MOSAIC_HIP_LIB=gpurun_out/tworcp.so python3 /tmp/explore.py >> $O/frontier.txt 2>&1
cat $O/frontier.txt
for rep in 1 2; do
  for c in D G; do
    for lib in "" gpurun_out/tworcp.so; do
      MOSAIC_HIP_LIB=$lib python3 bench.py --config $c --steps 8 --warmup 2 --no-cpu --opt pretier=1 > $O/x.json 2>/dev/null
      python3 - $O/x.json "$c pretier=1 lib=[$lib]" <<'PY'
import json, sys
d = json.load(open(sys.argv[1])); r = d["roofline"]
print(sys.argv[2], "ms/step %.3f" % d["ms_per_step"], "kernel %.3f" % r["avg_launch_ms"], "visited %.4f" % r["visited_fraction_of_groups"])
PY
    done
  done
done
