"""Photographs at scale: does the pre-tier (on by the size rule from 1e10 pixel pairs) pay on photographic content?
The photo fixture's target enlarged 2x and 3x (nearest neighbour: every cell's content stays photographic), its 1024 tiles as the
library, config D's parameters; pretier 0 / 1, ms per generate (best of 5)."""
import sys, time
import numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'tests'); sys.path.insert(0, 'tests/golden')
import cases, make_photo_fixture
from mosaicmagnifique_amd import GridUtility as G, synth
from mosaicmagnifique_amd.engine import Engine

img_u8, lib_u8 = make_photo_fixture.load()
e = Engine(0)
shape = cases.square_shape(64)
group = cases.group_of(shape)
lib = synth.preprocess(lib_u8, 2)
e.set_library(lib); e.set_masks(shape.masks4())
for f in (1, 2, 3):
    big = np.repeat(np.repeat(img_u8, f, axis=0), f, axis=1)
    img = synth.preprocess(big, 2)
    e.set_target(img)
    state = G.getGridState(group, img.shape[0], img.shape[1])[0]
    cells = G.cell_records(group, state, img.shape[0], img.shape[1])
    res = {}
    for pre in (0, 1, 0, 1):
        e.set_option("pretier", pre)
        best = None
        ts = []
        for _ in range(6):
            t0 = time.perf_counter(); b, s = e.generate(cells, state.shape[1], state.shape[0], 2, 4, 500); ts.append((time.perf_counter() - t0) * 1e3)
        tm = e.timing()
        res.setdefault(pre, []).append((min(ts[1:]), tm["visited_groups"] / max(tm["total_groups"], 1), tm["spec_rounds"], b))
    # the measured rule (option -1): launches 1 and 2 are the trial (off, on), from the third on the kept form
    e.set_option("pretier", -1)
    ts = []
    for _ in range(7):
        t0 = time.perf_counter(); b, s = e.generate(cells, state.shape[1], state.shape[0], 2, 4, 500); ts.append((time.perf_counter() - t0) * 1e3)
    res[-1] = [(min(ts[3:]), 0.0, 0, b)]
    kept = e.timing()["pretier"]
    same = all(np.array_equal(res[0][0][3], r[3]) for p in res for r in res[p])
    print("x%d: %d cells, %.3g pixel pairs | pretier 0: %s ms  pretier 1: %s ms | visited %.4f / %.4f | same grid %s" % (
        f, len(cells), len(cells) * 1024.0 * 4096, ["%.2f" % r[0] for r in res[0]], ["%.2f" % r[0] for r in res[1]], res[0][0][1], res[1][0][1], same),
        "| measured rule: %.2f ms, kept pretier=%d (launch times %s)" % (res[-1][0][0], kept, ["%.1f" % t for t in ts]), flush=True)
