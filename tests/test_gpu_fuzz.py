"""Seeded random scenarios through the whole path (u8 inputs -> device pre-processing -> detail
resize -> filter / re-score / resolve) against the oracle fed the same pre-processed arrays:
random cell sizes (odd ones too), random masks, spacings, offsets and flips, any detail level,
every colour difference, repeats on and off.  Each case is small; together they walk the
corners the hand-written cases do not (ragged last groups of the tile-major plane order,
2- and 4-block lanes of the early-exit bound, clipped cells at fractional detail ...)."""
import numpy as np
import pytest

import cases
from mosaicmagnifique_amd import GridUtility as G
from mosaicmagnifique_amd import HIPPhotomosaicGenerator, synth
from mosaicmagnifique_amd.cellshape import CellGroup, CellShape
from mosaicmagnifique_amd.colourdifference import Type
from oracle_ffi import shape_of

pytestmark = pytest.mark.gpu


def fuzz_seeds(name, default):
    """The seeds of one family: `default` of them, or MOSAIC_FUZZ_<NAME> many (the soaks), starting at MOSAIC_FUZZ_BASE
    (0: the scenarios every run of the suite sees; a soak with another base walks scenarios no run has seen before)."""
    import os
    base = int(os.environ.get("MOSAIC_FUZZ_BASE", "0"))
    return range(base, base + int(os.environ.get("MOSAIC_FUZZ_" + name, str(default))))


def random_shape(rng, size):
    kind = rng.integers(3)
    if kind == 0:
        return CellShape(size)
    yy, xx = np.mgrid[0:size, 0:size]
    if kind == 1:                                   # ellipse
        c = (size - 1) / 2.0
        mask = (((yy - c) / (c + 0.6)) ** 2 + ((xx - c) / (c * rng.uniform(0.6, 1.0) + 0.6)) ** 2 <= 1.0)
    else:                                           # random blobs (some rows / columns may be empty)
        mask = rng.uniform(size=(size, size)) < rng.uniform(0.3, 0.9)
        mask[size // 2, size // 2] = True
    s = CellShape(np.where(mask, 255, 0).astype(np.uint8))
    s.rowSpacing = int(rng.integers(max(1, size // 2), size + 1))
    s.colSpacing = int(rng.integers(max(1, size // 2), size + 1))
    s.alternateRowSpacing = s.rowSpacing if rng.uniform() < 0.5 else int(rng.integers(max(1, size // 2), size + 1))
    s.alternateColSpacing = s.colSpacing if rng.uniform() < 0.5 else int(rng.integers(max(1, size // 2), size + 1))
    s.alternateRowOffset = int(rng.integers(0, size // 2 + 1))
    s.alternateColOffset = int(rng.integers(0, size // 2 + 1))
    s.alternateColFlipHorizontal, s.alternateColFlipVertical, s.alternateRowFlipHorizontal, s.alternateRowFlipVertical = \
        (bool(v) for v in rng.integers(0, 2, 4))
    return s


@pytest.mark.parametrize("seed", fuzz_seeds("SMALL", 150))       # soak: MOSAIC_FUZZ_SMALL=1500
def test_random_scenario(oracle, seed):
    rng = np.random.default_rng(1000 + seed)
    size = int(rng.choice([8, 13, 16, 17, 24, 31, 32, 33, 40, 50, 64, 72]))
    shape = random_shape(rng, size)
    detail = int(rng.choice([100, 100, 50, 75, 60, 33, 25, 90]))
    t = Type(int(rng.integers(0, 3)))
    rep = (0, 0) if rng.uniform() < 0.35 else (int(rng.integers(1, 5)), int(rng.choice([1, 40, 700, 50000])))
    rows = int(rng.integers(size + 1, 4 * size + 7))
    cols = int(rng.integers(size + 1, 5 * size + 3))
    n_lib = int(rng.integers(3, 60))
    img = synth.target_u8(rows, cols, lattice=int(rng.choice([8, 32, 200])), seed=0x7A6 + seed)
    lib = synth.library_u8(n_lib, size, seed=0x11B + seed)
    if rng.uniform() < 0.3:                                        # exact duplicates: the lowest index must win
        lib[int(rng.integers(n_lib))] = lib[int(rng.integers(n_lib))]
    group = CellGroup(shape)
    group.setDetail(detail)
    state = G.getGridState(group, rows, cols)
    gen = HIPPhotomosaicGenerator(0)
    gen.setMainImage(img); gen.setLibrary(lib); gen.setColourDifference(t)
    gen.setCellGroup(group); gen.setGridState(state); gen.setRepeat(*rep)
    main, libf = gen.preprocessMainImage(), gen.preprocessLibraryImages()
    assert gen.generateBestFits(), gen.lastError
    d = group.getCell(0, True)
    want, want_score, _ = oracle.generate_step(shape_of(group.getCell(0)), d.getSize(), group.getDetail(), main, libf,
                                               d.masks4(), int(t), rep[0], rep[1], state=state[0])
    got, score = gen.getBestFits()[0], gen.getBestScores()[0]
    assert np.array_equal(got, want), (size, detail, int(t), rep, np.argwhere(got != want)[:4].tolist())
    valid = want >= 0
    tol = 1e-4 if t == Type.CIEDE2000 else 1e-12
    assert (np.abs(score[valid] - want_score[valid]) <= tol * np.abs(want_score[valid]) + 1e-300).all()


@pytest.mark.parametrize("seed", fuzz_seeds("STEPS", 40))
def test_random_scenario_with_size_steps(oracle, seed):
    """The same, over size steps (entropy subdivision on a half-busy image, library halved per step,
    cells resized per step) wherever the sizes stay consistent; checked step by step."""
    from test_gpu_detail_steps import run_steps
    rng = np.random.default_rng(5000 + seed)
    size = int(rng.choice([16, 24, 32, 40, 48, 64]))
    steps = int(rng.choice([1, 1, 2])) if size % 8 == 0 else 1
    detail = int(rng.choice([100, 50, 25])) if size % (4 << steps) == 0 else 100
    if size == 32 and rng.uniform() < 0.4:
        detail, steps = 75, 1                                     # 32 -> 24, 16 -> 12: general resize in both steps
    shape = random_shape(rng, size)
    # halving keeps spacings >= 1 and the step sizes whole only for even spacings: keep the random ones even
    for name in ("rowSpacing", "colSpacing", "alternateRowSpacing", "alternateColSpacing"):
        setattr(shape, name, max(2, getattr(shape, name) // 2 * 2))
    t = Type(int(rng.integers(0, 3)))
    rep = (0, 0) if rng.uniform() < 0.4 else (int(rng.integers(1, 4)), int(rng.choice([30, 900])))
    rows, cols = int(rng.integers(2 * size, 4 * size)), int(rng.integers(2 * size, 5 * size))
    img = synth.target_u8(rows, cols, lattice=int(rng.choice([16, 64])), seed=0x7A6 + seed)
    noise = rng.integers(-110, 111, (rows, cols // 2, 1))
    img[:, cols - cols // 2:] = np.clip(img[:, cols - cols // 2:].astype(np.int16) + noise, 0, 255).astype(np.uint8)   # busy half: splits
    lib = synth.library_u8(int(rng.integers(5, 50)), size, seed=0x11B + seed)
    run_steps(oracle, shape, img, lib, t, detail, steps, rep)


@pytest.mark.parametrize("seed", fuzz_seeds("LARGE", 24))     # soak: MOSAIC_FUZZ_LARGE=300
def test_random_scenario_large_library(oracle, engine, seed):
    """Scenarios the small ones above cannot reach: hundreds of library images (so most candidates are NOT seeds
    and go through the two stages of the bound, the bound-ordered visits and the early exit), 48 / 56 / 64 px cells with
    random masks, and at least 512 cells (so the persistent workgroups split the cells into per-XCD shares).
    Every cell is checked by the oracle as the reference's raster loop reaches it, given the engine's earlier
    picks (oracle_ffi.picks_given_grid: equality on the whole raster = equality with the sequential run)."""
    import os
    rng = np.random.default_rng(9000 + seed)
    size = int(rng.choice([48, 56, 64]))
    shape = random_shape(rng, size)
    t = Type.CIEDE2000 if seed % 3 != 2 else Type.CIE76
    rep = (0, 0) if seed % 2 else (int(rng.integers(1, 5)), int(rng.choice([40, 700])))
    n_lib = int(rng.integers(150, 400))
    rows = cols = 23 * max(shape.rowSpacing, shape.colSpacing, shape.alternateRowSpacing, shape.alternateColSpacing) + int(rng.integers(0, size))
    img = synth.preprocess(synth.target_u8(rows, cols, lattice=int(rng.choice([32, 200])), seed=0x7A6 + seed), t)
    lib = synth.preprocess(synth.library_u8(n_lib, size, seed=0x11B + seed), t)
    group = CellGroup(shape)
    state = G.getGridState(group, rows, cols)[0]
    cells = G.cell_records(group, state, rows, cols)
    assert len(cells) >= 512
    engine.set_library(lib); engine.set_masks(shape.masks4()); engine.set_target(img)
    best, score = engine.generate(cells, state.shape[1], state.shape[0], int(t), *rep)
    tm = engine.timing()
    if t == Type.CIEDE2000:
        assert tm["visited_groups"] > 0 and tm["band_alarms"] == 0
    # no launch voided (the pre-tier's canary, the audit of the dropped candidates: a sticky word of the filter's counter block, counted
    # per launch) -- with MOSAIC_TEST_OPTS=pretier=1,audit=1 this says that EVERY drop of the scenario was sound
    assert tm["voided_launches"] == 0, tm
    grid = np.where(best < n_lib, best, -1)
    pos = np.stack([cells["grid_y"], cells["grid_x"]], axis=1)
    threads = min(16, len(os.sched_getaffinity(0)))
    picks, scores = oracle.picks_given_grid(shape_of(shape), img, lib, shape.masks4(), int(t), rep[0], rep[1], grid, pos,
                                            threads=threads)
    got = best[pos[:, 0], pos[:, 1]]
    assert np.array_equal(np.where(got == n_lib, -1, got), picks), np.flatnonzero(np.where(got == n_lib, -1, got) != picks)[:5]
    ok = picks >= 0
    tol = 1e-4 if t == Type.CIEDE2000 else 1e-12
    assert (np.abs(score[pos[ok, 0], pos[ok, 1]] - scores[ok]) <= tol * np.abs(scores[ok]) + 1e-300).all()


OPTION_SETS = [
    {"interleave": 0},                              # planar images: filter_kernel<.., IL = false> (raw operands of the fp32 form)
    {"xcd_aware": 0},                               # one item counter for the whole launch
    {"tile": 16, "xcd_cells": 16},                  # small items, small XCD sub-groups
    {"tile": 64, "xcd_cells": 200},                 # the largest tile of the two-stage form, ragged sub-groups
    {"tile": 128},                                  # tiles above 64 candidates: single-stage bound
    {"two_level": 0, "sort_groups": 0},             # per-candidate bound, plane-order visits
    {"lds_filter": 1, "tile": 64},                  # the LDS-resident cell tile, interleaved operands
    {"lds_filter": 1, "interleave": 0},             # ... and planar ones
    {"seeds": 0},                                   # no extra seeds: the limit rests on K proxies only
    {"front_resolve": 0},                           # further sets of sweeps / round 1's row-pipelined kernel instead of the front form of the wavefront
    {"sweep_resolve": 0},                           # the front form from the start (step-by-step path)
    {"spec_k": 2},                                  # short speculative lists: cells get deepened (from the result block's verdicts)
    {"spec_k": 2, "deepen_mid": 0},                 # ... straight to the full counting length, as before round 6
    {"spec_k": 2, "warm_rounds": 0},                # ... every round's resolve from a cleared grid (default: a later round's sweeps start from the previous round's fix point)
    {"spec_k": 2, "deepen_mid": 5},                 # ... through a middle length of 5 entries first (three levels: 2 -> 5 -> full)
    {"spec_auto": 0, "spec_k": 1, "deepen_mid": 3, "fast_path": 0},
    {"fused": 0},                                   # separate gather / chroma / interleave / statistics and contenders / re-score / re-rank launches
    {"fast_path": 0},                               # the step-by-step path: the host decides between the phases
    {"fast_path": 0, "fused": 0, "spec_k": 2},      # ... as round 2 ran it, with deepening rounds
    {"item_auto": 0},                               # item size pinned to the by-cell-size default
    {"stage0": 0},                                  # no whole-image bound ahead of the block bound
    {"euclid_stage1": 0},                           # (RGB / CIE76 only; a no-op for this CIEDE2000 scenario)
    {"fused_seeds": 0},                             # seeds, their scores and the limit as three launches
    {"prune_min_kpairs": -1, "phase_timing": -1},   # the engine's own defaults: no seeds / early exit and no phase events below a work threshold
    {"prune_min_kpairs": 2000000000},               # never pruned
    {"pretier": 1},                                 # the packed-fp16 pre-tier ahead of the fp32 filter form (default: measured on cells above 32 groups)
    {"pretier": 1, "pretier_band": 1},              # ... on round 4's observed band instead of the worst-case one (the audit eight times as dense)
    {"pretier": 0},
    {"pretier": 1, "tile": 64},
    {"pretier": 1, "audit": 1},                     # ... with EVERY candidate that leaves without a full score walked again without the bound (default: one in 1024)
    {"audit": 1},
    {"pretier": 1, "audit": 0},
    {"spec_auto": 0, "spec_k": 3},                  # the short-list length pinned
    {"cell_order": 0},                              # the cells of an XCD's share in their given order (default: clipped cells first)
    {"cell_order": 0, "xcd_cells": 16, "tile": 16}, # ... with small sub-groups
]


@pytest.mark.gpu
@pytest.mark.parametrize("opts", OPTION_SETS, ids=lambda o: ",".join("%s=%s" % kv for kv in o.items()))
def test_tuning_options_do_not_change_the_result(oracle, engine, opts):
    """Every tuning option of the filter (mm_set_option) is a pure performance knob: on a scenario with >= 512 cells
    (per-XCD shares), a ragged hexagon-ish mask, 260 images and a repeat penalty the grid and the fp64 scores equal the
    default's bit for bit; the default's grid is checked against the oracle cell by cell."""
    import os
    rng = np.random.default_rng(4242)
    size = 48
    shape = random_shape(rng, size)
    t, rep, n_lib = Type.CIEDE2000, (2, 300), 260
    rows = cols = 23 * max(shape.rowSpacing, shape.colSpacing, shape.alternateRowSpacing, shape.alternateColSpacing) + 17
    img = synth.preprocess(synth.target_u8(rows, cols, lattice=64, seed=0x7A6 + 77), t)
    lib = synth.preprocess(synth.library_u8(n_lib, size, seed=0x11B + 77), t)
    group = CellGroup(shape)
    state = G.getGridState(group, rows, cols)[0]
    cells = G.cell_records(group, state, rows, cols)
    assert len(cells) >= 512
    engine.set_library(lib); engine.set_masks(shape.masks4()); engine.set_target(img)
    best, score = engine.generate(cells, state.shape[1], state.shape[0], int(t), *rep)
    defaults = {"interleave": 1, "xcd_aware": 1, "tile": 0, "xcd_cells": 0, "two_level": 1, "sort_groups": -1,
                "lds_filter": -1, "seeds": -1, "spec_k": 4, "fused": 1, "fast_path": 1, "item_auto": 1,
                "fused_seeds": 2048, "prune_min_kpairs": 0, "phase_timing": 1, "stage0": 1, "euclid_stage1": 1, "cell_order": 1,
                "pretier": -1, "pretier_band": 0, "spec_auto": 1, "audit": 1024, "front_resolve": 1, "sweep_resolve": 1, "deepen_mid": 16, "warm_rounds": 1}
    try:
        for name, value in opts.items():
            engine.set_option(name, value)
        got, got_score = engine.generate(cells, state.shape[1], state.shape[0], int(t), *rep)
        tm = engine.timing()
    finally:
        for name in opts:
            engine.set_option(name, defaults[name])
    assert np.array_equal(got, best) and np.array_equal(got_score[state >= 0], score[state >= 0])
    assert tm["band_alarms"] == 0 and tm["voided_launches"] == 0
    if opts.get("fast_path") == 0:
        assert tm["fast_path"] == 0 and tm["host_syncs"] >= 3
    elif "spec_k" not in opts:
        assert tm["host_syncs"] >= 1
    if opts == OPTION_SETS[0]:                      # the oracle once
        grid = np.where(best < n_lib, best, -1)
        pos = np.stack([cells["grid_y"], cells["grid_x"]], axis=1)
        picks, _ = oracle.picks_given_grid(shape_of(shape), img, lib, shape.masks4(), int(t), rep[0], rep[1], grid, pos,
                                           threads=min(16, len(os.sched_getaffinity(0))))
        mine = best[pos[:, 0], pos[:, 1]]
        assert np.array_equal(np.where(mine == n_lib, -1, mine), picks)


@pytest.mark.gpu
@pytest.mark.parametrize("seed", fuzz_seeds("SMOOTH", 16))
def test_random_scenario_graded_noise(oracle, engine, seed):
    """Between the synthetic library's +-32 of per-channel noise and noise-free gradients: library images = a base colour
    + a linear gradient + noise of a random amplitude (0 .. 24), target = smooth waves + noise (0 .. 12).  The bound then
    drops anything from half to nearly all of the candidates, and the engine's item size follows (64-candidate items
    after a launch that dropped >= 90 %): every cell against the oracle, given the engine's earlier picks."""
    import os
    rng = np.random.default_rng(31000 + seed)
    size = int(rng.choice([32, 48, 64]))
    shape = random_shape(rng, size)
    t = Type.CIEDE2000
    rep = (0, 0) if seed % 2 else (int(rng.integers(1, 4)), int(rng.choice([30, 400])))
    n_lib = int(rng.integers(150, 320))
    amp_lib, amp_img = float(rng.choice([0, 2, 8, 24])), float(rng.choice([0, 3, 12]))
    yy, xx = np.mgrid[0:size, 0:size].astype(np.float64) / size
    base = rng.uniform(15, 240, (n_lib, 1, 1, 3))
    gy, gx = rng.uniform(-45, 45, (n_lib, 1, 1, 3)), rng.uniform(-45, 45, (n_lib, 1, 1, 3))
    lib_u8 = base + yy[None, ..., None] * gy + xx[None, ..., None] * gx + rng.uniform(-amp_lib, amp_lib, (n_lib, size, size, 3))
    lib_u8 = np.clip(lib_u8, 0, 255).round().astype(np.uint8)
    rows = cols = 23 * max(shape.rowSpacing, shape.colSpacing, shape.alternateRowSpacing, shape.alternateColSpacing) + int(rng.integers(0, size))
    Y, X = np.mgrid[0:rows, 0:cols].astype(np.float64)
    img_u8 = np.stack([127 + 105 * np.sin(X / rng.uniform(90, 400) + c) * np.cos(Y / rng.uniform(90, 400) - c) for c in (0.0, 1.1, 2.3)], -1)
    img_u8 = np.clip(img_u8 + rng.uniform(-amp_img, amp_img, img_u8.shape), 0, 255).round().astype(np.uint8)
    img, lib = synth.preprocess(img_u8, t), synth.preprocess(lib_u8, t)
    group = CellGroup(shape)
    state = G.getGridState(group, rows, cols)[0]
    cells = G.cell_records(group, state, rows, cols)
    engine.set_library(lib); engine.set_masks(shape.masks4()); engine.set_target(img)
    for call in range(2):                                   # the second call runs with the item size the first one's drop rate picks
        best, score = engine.generate(cells, state.shape[1], state.shape[0], int(t), *rep)
        tm = engine.timing()
        assert tm["band_alarms"] == 0 and tm["voided_launches"] == 0, tm       # (no launch voided: see test_random_scenario_large_library)
        if call == 0:
            first = (best.copy(), score.copy())
    assert np.array_equal(best, first[0]) and np.array_equal(score[state >= 0], first[1][state >= 0])
    grid = np.where(best < n_lib, best, -1)
    pos = np.stack([cells["grid_y"], cells["grid_x"]], axis=1)
    picks, scores = oracle.picks_given_grid(shape_of(shape), img, lib, shape.masks4(), int(t), rep[0], rep[1], grid, pos,
                                            threads=min(16, len(os.sched_getaffinity(0))))
    got = best[pos[:, 0], pos[:, 1]]
    assert np.array_equal(np.where(got == n_lib, -1, got), picks), np.flatnonzero(np.where(got == n_lib, -1, got) != picks)[:5]
