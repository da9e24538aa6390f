"""Work counters of the filter kernel for one generate of a bench configuration (always on since round 2:
mm_timing.visited_groups / total_groups / bound_dropped).   python tools/visit_counters.py [D|C|E8] [NAME=VALUE ...]"""
import os
import sys

sys.path.insert(0, os.getcwd())
sys.path.insert(0, os.path.join(os.getcwd(), 'tests'))
config = sys.argv[1] if len(sys.argv) > 1 else 'D'
opts = sys.argv[2:]
sys.argv = ['bench.py']
import bench  # noqa: E402
from mosaicmagnifique_amd.engine import Engine  # noqa: E402

w = bench.build_workload(config, 1)
e = Engine(0)
for item in opts:
    name, value = item.split('=')
    e.set_option(name, int(value))
bench.stage(e, w)
e.generate(w["cells"], w["grid_w"], w["grid_h"], w["t"], *w["rep"])
t = e.timing()
n = len(w['cells']) * w['L']
print(t)
print('visited groups %d of %d = %.3f ; candidates dropped by the bound %d of %d = %.3f' % (
    t['visited_groups'], t['total_groups'], t['visited_groups'] / max(t['total_groups'], 1), t['bound_dropped'], n, t['bound_dropped'] / n))
if os.environ.get("MOSAIC_HIP_LIB"):        # an instrumented build prints its cycle accounts from here (stderr)
    import numpy as np
    from mosaicmagnifique_amd import _lib
    raw = np.zeros(3, np.uint64)
    _lib.load().mm_debug_visit_counters(e._h, _lib.ptr(raw))
