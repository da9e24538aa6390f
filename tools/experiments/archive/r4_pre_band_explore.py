import sys
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from mosaicmagnifique_amd.engine import Engine
e = Engine(0)
grid = 512 * 512 * 24 * 24 * 25
for rel6 in (0, 1000, 2000, 3000, 4000, 6000, 8000):
    for abs4 in (300, 600, 1000, 1500, 2500, 3500):
        e.set_option("dbg_pre_abs_1e4", abs4); e.set_option("dbg_pre_rel_1e6", max(rel6, 1) if abs4 == 0 else rel6 if rel6 else 1)
        r = e.filter_band_sweep(2, grid // 4, seed=1)
        r3 = e.filter_band_sweep(3, 512 * 128 * 24 * 24 * 25 // 4, seed=7)
        print("abs %.3f rel %.4f  share mode2 %.3f mode3 %.3f  max_err %.3f" % (abs4 * 1e-4, rel6 * 1e-6, r["band_share"], r3["band_share"], r["max_err"]), flush=True)
