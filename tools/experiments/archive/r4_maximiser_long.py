"""Round 4: the adversarial maximiser of tests/test_gpu_filter.py with more seeds and generations than the suite affords,
on all three forms of the CIEDE2000 filter (0: fp32, 1: fp32 on prepared operands, 2: the packed-fp16 pre-tier).
usage (GPU box): python3 tools/experiments/r4_maximiser_long.py [seeds] [generations]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import test_gpu_filter as T  # noqa: E402
from mosaicmagnifique_amd.engine import Engine  # noqa: E402

seeds = int(sys.argv[1]) if len(sys.argv) > 1 else 12
gens = int(sys.argv[2]) if len(sys.argv) > 2 else 400
with Engine(0) as engine:
    for prepared in (2, 0, 1):
        worst = 0.0
        for seed in range(100, 100 + (seeds if prepared == 2 else max(2, seeds // 4))):
            t0 = time.time()
            start, best, a, b, exact = T.climb(engine, prepared, seed, gens)
            worst = max(worst, best)
            print("form %d seed %d: %.3f -> %.3f  %s vs %s  dE00 %.4f  (%.0f s)" % (prepared, seed, start, best, a, b, exact, time.time() - t0), flush=True)
        print("form %d: worst share over the seeds %.3f" % (prepared, worst), flush=True)
