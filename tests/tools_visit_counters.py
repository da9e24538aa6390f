import sys, os, ctypes, numpy as np
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), 'tests'))
config = sys.argv[1] if len(sys.argv) > 1 else 'D'
sys.argv=['bench.py']
import bench, torch
from mosaicmagnifique_amd.engine import Engine
w = bench.build_workload(config, 1)
e = Engine(0); e.set_library(w["lib"]); e.set_masks(w["shape"].masks4()); e.set_target(w["img"])
best, score = e.generate(w["cells"], w["grid_w"], w["grid_h"], w["t"], *w["rep"])
print(e.timing())
out=(ctypes.c_ulonglong*3)()
e._lib.mm_debug_visit_counters(e._h, out)
v,t,lb=out[0],out[1],out[2]
print('visited groups %d of %d = %.3f ; candidates dropped by lightness bound %d of %d = %.3f' % (v,t,v/max(t,1),lb,len(w['cells'])*w['L'],lb/(len(w['cells'])*w['L'])))
