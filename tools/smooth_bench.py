"""One-off measurement: config D's shape (4096 x 4096 target, 64 x 64 cells, 5000 images, CIEDE2000, repeat 4 / 500)
on NOISE-FREE content -- smooth gradients on both sides, as photographs are next to the synthetic +-32 library --
to see what the early-exit bound does there.  python tools/smooth_bench.py (on the GPU box)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import cases
from mosaicmagnifique_amd import GridUtility as G
from mosaicmagnifique_amd.engine import Engine

size, n_lib, rows, cols, t = 64, 5000, 4096, 4096, 2
rng = np.random.default_rng(5)
yy, xx = np.mgrid[0:size, 0:size].astype(np.float32) / size
base = rng.uniform(10, 245, (n_lib, 1, 1, 3)).astype(np.float32)
gy = rng.uniform(-50, 50, (n_lib, 1, 1, 3)).astype(np.float32)
gx = rng.uniform(-50, 50, (n_lib, 1, 1, 3)).astype(np.float32)
lib_u8 = np.clip(base + yy[None, ..., None] * gy + xx[None, ..., None] * gx, 0, 255).round().astype(np.uint8)
Y, X = np.mgrid[0:rows, 0:cols].astype(np.float32)
img_u8 = np.stack([127 + 110 * np.sin(X / 310 + c) * np.cos(Y / 270 - c) + 15 * np.sin(X / 23 + Y / 31 + c) for c in (0.0, 1.1, 2.3)], -1)
img_u8 = np.clip(img_u8, 0, 255).round().astype(np.uint8)
shape = cases.square_shape(size)
group = cases.group_of(shape)
state = G.getGridState(group, rows, cols)[0]
cells = G.cell_records(group, state, rows, cols)
e = Engine(0)
e.set_library_u8(lib_u8, t); e.set_target_u8(img_u8, t); e.set_masks(shape.masks4())
for name, opts in (("default", {}), ("stage0=0", {"stage0": 0}), ("tile=64", {"tile": 64}), ("tile=16", {"tile": 16}), ("two_level=0", {"two_level": 0}), ("prune=0", {"prune": 0})):
    for k, v in opts.items():
        e.set_option(k, v)
    for i in range(3):
        t0 = time.perf_counter()
        best, score = e.generate(cells, state.shape[1], state.shape[0], t, 4, 500)
        dt = time.perf_counter() - t0
    tm = e.timing()
    print(name, "step %.2f ms, %.0f cells/s, filter %.2f ms, seeds %.2f, rescore %.2f, visited %.4f of groups, dropped by the bound %.3f of candidates, alarms %d"
          % (dt * 1e3, len(cells) / dt, tm["score_ms"], tm["seed_ms"], tm["rescore_ms"],
             tm["visited_groups"] / max(tm["total_groups"], 1), tm["bound_dropped"] / (len(cells) * n_lib), tm["band_alarms"]))
    if name == "default":
        ref = (best.copy(), score.copy())
    else:
        print("same as default:", bool(np.array_equal(best, ref[0]) and np.array_equal(score[state >= 0], ref[1][state >= 0])))
    for k in opts:
        e.set_option(k, {"tile": 0}.get(k, 1))
