# Round 4: the pre-tier decided by measurement (first launch off, second on, the faster kept).  The whole suite, photographs at
# scale, the benchmark configurations.
set -x
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_auto; mkdir -p $O
timeout -k 10 1000 python3 -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log; tail -5 $O/pytest.log | cut -c1-300
python3 tools/experiments/r4_photo_scale.py > $O/photo_scale.txt 2>&1; tail -4 $O/photo_scale.txt
for c in D E8 G I C P; do python3 bench.py --config $c --steps 8 --warmup 3 --no-cpu > $O/x.json 2>/dev/null; python3 - $O/x.json $c <<'PY'
import json, sys
d = json.load(open(sys.argv[1])); t = d["last_step_device_ms"]
print("bench", sys.argv[2], "ms/step %.3f" % d["ms_per_step"], "kernel %.3f" % d["roofline"]["avg_launch_ms"], "pretier kept", t["pretier"], d["dtype"])
PY
done
