"""Times the dominant kernel of one bench config under the engine's options
(diagnostics; not a test).  usage: python tools/filter_variants.py [CONFIG]"""
import os
import sys

sys.path.insert(0, os.getcwd())
sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
config = sys.argv[1] if len(sys.argv) > 1 else "D"
sys.argv = ["bench.py"]
import bench  # noqa: E402
from mosaicmagnifique_amd.engine import Engine  # noqa: E402

w = bench.build_workload(config, 1)
e = Engine(0)
e.set_library(w["lib"]); e.set_masks(w["shape"].masks4()); e.set_target(w["img"])
for name, opts in [("default", {}), ("tile=8", {"tile": 8}), ("tile=16", {"tile": 16}), ("tile=32", {"tile": 32}),
                   ("tile=64", {"tile": 64}), ("tile=128", {"tile": 128}), ("prune=0", {"prune": 0}), ("lds_filter=1", {"lds_filter": 1}),
                   ("lds_filter=1 prune=0", {"lds_filter": 1, "prune": 0}), ("filter=0 (fp64 only)", {"filter": 0})]:
    if name.startswith("filter=0") and config in ("D", "E8"):
        continue
    for k in ("prune", "filter"):
        e.set_option(k, 1)
    e.set_option("lds_filter", -1)
    e.set_option("tile", 0)
    for k, v in opts.items():
        e.set_option(k, v)
    for _ in range(2):
        e.generate(w["cells"], w["grid_w"], w["grid_h"], w["t"], *w["rep"])
    t = e.timing()
    print("%-24s score_ms=%8.3f seed_ms=%6.3f rescore_ms=%6.3f total_ms=%8.3f" % (name, t["score_ms"], t["seed_ms"], t["rescore_ms"], t["total_ms"]))
