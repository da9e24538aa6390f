"""generateBestFits through the C ABI against the oracle: the reference's
CPUvsCUDA COMPARE_* matrix (test/tst_CUDAGenerator.h:530-815 -- CPU best-fit grid ==
device best-fit grid, cell by cell) on seeded inputs, plus the edge cases and
error behaviour of the boundary."""
import ctypes

import numpy as np
import pytest

import cases
from mosaicmagnifique_amd import GridUtility as G
from mosaicmagnifique_amd import HIPPhotomosaicGenerator
from mosaicmagnifique_amd.colourdifference import Type
from mosaicmagnifique_amd.engine import Engine, EngineError
from oracle_ffi import shape_of

pytestmark = pytest.mark.gpu

SHAPES = {
    "square16": lambda: cases.square_shape(16),
    "hexagon64": cases.hexagon_shape,
    "flippy24": cases.flippy_shape,
}
# score tolerance (relative): Euclid terms are bit-exact, only the summation order
# differs; CIEDE2000 is the contract's 1e-4 (observed ~1e-13).
REL_TOL = {0: 1e-12, 1: 1e-12, 2: 1e-4}


# (wave_resolve, row_pipeline, sweep_resolve, front_resolve): the parallel sweeps (default; content they do not settle goes on to the
# front form), the wavefront in its latency-built front form (round 6) and in round 1's row-pipelined form, the single-wave chain,
# the workgroup-per-cell chain
RESOLVE_MODES = {"sweep": (1, 1, 1, 1), "rows": (1, 1, 0, 1), "rows_r1": (1, 1, 0, 0), "sweep_r5": (1, 1, 1, 0), "wave": (1, 0, 0, 1),
                 "workgroup": (0, 0, 0, 1)}


def generate(shape, img, lib, t, rep=(0, 0), device=0, use_filter=1, wave_resolve="sweep", options=()):
    group = cases.group_of(shape)
    gen = HIPPhotomosaicGenerator(device)
    gen.m_engine.set_option("filter", use_filter)
    gen.m_engine.set_option("wave_resolve", RESOLVE_MODES[wave_resolve][0])
    gen.m_engine.set_option("row_pipeline", RESOLVE_MODES[wave_resolve][1])
    gen.m_engine.set_option("sweep_resolve", RESOLVE_MODES[wave_resolve][2])
    gen.m_engine.set_option("front_resolve", RESOLVE_MODES[wave_resolve][3])
    for name, value in options:
        gen.m_engine.set_option(name, value)
    gen.setMainImage(img)
    gen.setLibrary(lib)
    gen.setColourDifference(t)
    gen.setCellGroup(group)
    gen.setGridState(G.getGridState(group, img.shape[0], img.shape[1]))
    gen.setRepeat(*rep)
    assert gen.generateBestFits(), gen.lastError
    return gen


def compare(oracle, shape, img, lib, t, rep=(0, 0), use_filter=1, wave_resolve="sweep", options=()):
    gen = generate(shape, img, lib, t, rep, use_filter=use_filter, wave_resolve=wave_resolve, options=options)
    got, score = gen.getBestFits()[0], gen.getBestScores()[0]
    want, want_score, _ = oracle.generate_best_fits(shape_of(shape), img, lib, shape.masks4(), int(t), *rep)
    assert got.shape == want.shape
    mism = np.argwhere(got != want)
    assert len(mism) == 0, "first mismatches (gy,gx): %s got %s want %s" % (
        mism[:5].tolist(), got[tuple(mism[:5].T)].tolist(), want[tuple(mism[:5].T)].tolist())
    valid = want >= 0
    rel = np.abs(score[valid] - want_score[valid]) / np.maximum(np.abs(want_score[valid]), 1e-300)
    assert rel.max() <= REL_TOL[int(t)], rel.max()
    assert np.isnan(score[~valid]).all()
    return gen


# the engine as it SHIPS: no events between the phases of a small job, seeds / early exit only above the work threshold
# (tests/conftest.py turns both on for the suite's engines so that the tests can read which path ran)
SHIPPED_DEFAULTS = (("phase_timing", -1), ("prune_min_kpairs", -1))


@pytest.mark.parametrize("use_filter", [1, 0, "defaults"], ids=["two_tier", "fp64_only", "shipped_defaults"])
@pytest.mark.parametrize("rep", [(0, 0), (2, 300), (20, 10000)], ids=["norepeat", "repeat2", "repeat20"])
@pytest.mark.parametrize("shape_name", SHAPES)
@pytest.mark.parametrize("t", [Type.RGB_EUCLIDEAN, Type.CIE76, Type.CIEDE2000])
def test_compare_matrix(oracle, t, shape_name, rep, use_filter):
    """COMPARE_{RGB_EUCLIDEAN,CIE76,CIEDE2000} x {square, Hexagon, flipped/offset}
    x {no repeats, repeats, the reference's (20, 10000) repeats case}.  Ragged image
    sizes so the last row / column of cells (and the x=-1 / y=-1 cells of offset
    shapes) are clipped.  Each on the suite's engine options (every phase forced on), on the fp64-only path and on
    the options the engine ships with (what a small production job actually runs)."""
    shape = SHAPES[shape_name]()
    size = shape.getSize()
    rows, cols = 5 * size + size // 3, 6 * size + size // 2
    n_lib = 64 if size >= 64 else 150
    img, lib = cases.inputs(rows, cols, size, n_lib, t, lattice=2 * size)
    if use_filter == "defaults":
        gen = compare(oracle, shape, img, lib, t, rep, 1, options=SHIPPED_DEFAULTS)
        assert gen.timing()["band_alarms"] == 0
        return
    gen = compare(oracle, shape, img, lib, t, rep, use_filter)
    tm = gen.timing()
    assert (tm["rescore_ms"] > 0) == bool(use_filter and gen.m_engine.speculative_candidates(*rep) < n_lib)


@pytest.mark.parametrize("t", [Type.RGB_EUCLIDEAN, Type.CIEDE2000])
def test_sweep_resolve_reaches_the_raster_order_result(oracle, t):
    """The repeat penalty as parallel sweeps to the fix point (resolve_wave_kernel mode 2):
    on a nearly flat target every cell wants the same few images, so picks keep displacing
    each other along the raster order -- the longest dependency chains the path can see.
    Same grid and scores as the oracle's sequential scan with (a) enough sweeps, (b) a
    sweep budget too small to converge (falls back to the chain), (c) sweeps off."""
    size = 16
    shape = cases.square_shape(size)
    img, lib = cases.inputs(12 * size, 14 * size, size, 90, t, lattice=64 * size)   # one lattice cell: smooth, near-flat
    sweeps = {}
    for name, options in (("free", (("max_sweeps", 512),)), ("starved", (("max_sweeps", 2),)), ("off", (("sweep_resolve", 0),)),
                          ("free", (("max_sweeps", 512), ("front_resolve", 0)))):
        for rep in ((3, 250), (4, 100000), (2, -40)):
            gen = compare(oracle, shape, img, lib, t, rep=rep, options=options)
            sweeps[name, rep] = gen.timing()["resolve_sweeps"]
    assert all(sweeps["off", rep] == 0 for rep in ((3, 250), (4, 100000), (2, -40)))
    assert all(sweeps["starved", rep] >= 2 for rep in ((3, 250), (4, 100000), (2, -40)))     # 2 per resolve pass, then the chain
    assert all(2 <= sweeps["free", rep] <= 512 for rep in ((3, 250), (4, 100000), (2, -40))), sweeps
    assert max(sweeps["free", rep] for rep in ((3, 250), (4, 100000))) > 3, sweeps           # the chains were really long


@pytest.mark.parametrize("wave_resolve", list(RESOLVE_MODES))
@pytest.mark.parametrize("t", [Type.RGB_EUCLIDEAN, Type.CIEDE2000])
def test_duplicate_library_images_lowest_index_wins(oracle, t, wave_resolve):
    shape = cases.square_shape(16)
    img, lib = cases.inputs(96, 96, 16, 60, t)
    first = generate(shape, img, lib, t).getBestFits()[0]
    winners = np.unique(first[first >= 0])
    dups = [((int(w) + 31) % 60, int(w)) for w in winners[:6]]
    img, lib = cases.inputs(96, 96, 16, 60, t, duplicates=dups)
    compare(oracle, shape, img, lib, t, wave_resolve=wave_resolve)
    compare(oracle, shape, img, lib, t, rep=(1, 40), wave_resolve=wave_resolve)
    compare(oracle, shape, img, lib, t, rep=(9, 7), wave_resolve=wave_resolve)     # window > 64 neighbours


@pytest.mark.parametrize("wave_resolve", list(RESOLVE_MODES))
@pytest.mark.parametrize("t", [Type.RGB_EUCLIDEAN, Type.CIE76])
def test_near_tie_is_broken_in_reference_summation_order(oracle, t, wave_resolve):
    """Flat target + two library images that are pixel PERMUTATIONS of each other:
    the same multiset of per-pixel terms, so the two sums agree to ~1e-16 relative and
    only the reference's sequential order decides.  Forces the exact re-sum path."""
    size, n_lib = 16, 40
    shape = cases.square_shape(size)
    rng = np.random.default_rng(21)
    img, lib = cases.inputs(64, 64, size, n_lib, t)
    flat = img[0, 0].copy()
    img[:] = flat
    patch = (flat + rng.uniform(-3, 3, (size, size, 3))).astype(np.float32)
    if t == Type.RGB_EUCLIDEAN:
        patch = np.clip(np.rint(patch), 0, 255).astype(np.float32)
    for k, idx in enumerate((7, 3, 29)):
        perm = rng.permutation(size * size)
        lib[idx] = patch.reshape(-1, 3)[perm].reshape(size, size, 3) if k else patch
    gen = compare(oracle, shape, img, lib, t, wave_resolve=wave_resolve)
    tm = gen.timing()
    # 3 near-tied candidates but a 2-entry list: the list is marked as overflowing and
    # the cell's whole row is recomputed before the exact re-sum
    assert tm["exact_rescored"] >= 3 and tm["dense_rows"] > 0
    gen = compare(oracle, shape, img, lib, t, rep=(1, 1), wave_resolve=wave_resolve)
    assert gen.timing()["exact_rescored"] > 0 and gen.timing()["dense_rows"] == 0


def test_image_smaller_than_a_cell_and_single_pixel_bounds(oracle):
    shape = cases.square_shape(16)
    for rows, cols in [(10, 10), (1, 40), (17, 16)]:
        img, lib = cases.inputs(rows, cols, 16, 20, Type.CIEDE2000)
        compare(oracle, shape, img, lib, Type.CIEDE2000)


def test_edge_kernel_semantics_last_quarter_rows_ignored(oracle, engine):
    """*_CUDAEdgeCase (tst_ColourDifference.h:472-543): with target_area =
    {0, size - size/4, 0, size} the last quarter of the rows contributes nothing."""
    size = 16
    img, lib = cases.inputs(size, size, size, 12, Type.RGB_EUCLIDEAN)
    rec = np.zeros(1, dtype=G.CELL_DTYPE)
    rec["row_max"], rec["col_max"] = size - size // 4, size
    rec["grid_x"] = rec["grid_y"] = 2
    engine.set_library(lib)
    engine.set_masks(cases.square_shape(size).masks4())
    engine.set_target(img)
    best, score = engine.generate(rec, 5, 5, 0)
    img2 = img.copy()
    img2[size - size // 4:] = 255 - img2[size - size // 4:]      # scribble over the ignored rows
    engine.set_target(img2)
    best2, score2 = engine.generate(rec, 5, 5, 0)
    assert best[2, 2] == best2[2, 2] and score[2, 2] == score2[2, 2]
    # and equals the oracle's sum over rows [0, 12)
    d = oracle.colour_diff_batch(0, np.repeat(img[None, :12], 12, 0), lib[:, :12]).reshape(12, -1).sum(1)
    assert best[2, 2] == int(np.argmin(d))
    assert abs(score[2, 2] - d.min()) <= 1e-12 * d.min()


def test_result_layout_sentinel_and_writeback(oracle, engine):
    """Flat padded row-major grid with sentinel n_images for untouched cells
    (CUDAPhotomosaicGenerator.cpp:140-149,286,320); the generator writes back only
    where the grid state has a value."""
    shape = cases.square_shape(16)
    group = cases.group_of(shape)
    img, lib = cases.inputs(48, 64, 16, 25, Type.RGB_EUCLIDEAN)
    state = G.getGridState(group, 48, 64)[0]
    cells = G.cell_records(group, state, 48, 64)
    engine.set_library(lib)
    engine.set_masks(shape.masks4())
    engine.set_target(img)
    best, score = engine.generate(cells, state.shape[1], state.shape[0], 0)
    assert best.shape == state.shape == (5, 6)
    assert (best[state < 0] == 25).all() and (best[state >= 0] < 25).all()
    assert np.isnan(score[state < 0]).all()
    # a caller that masks cells out (size-step holes): those stay nullopt
    state2 = state.copy()
    state2[3, 3] = G.NULLOPT
    gen = HIPPhotomosaicGenerator(0)
    gen.setMainImage(img); gen.setLibrary(lib); gen.setCellGroup(group)
    gen.setGridState([state2]); gen.setColourDifference(Type.RGB_EUCLIDEAN)
    assert gen.generateBestFits()
    out = gen.getBestFits()[0]
    assert out[3, 3] == G.NULLOPT and (out[state2 >= 0] == best[state2 >= 0]).all()


def test_consistency_and_progress(oracle):
    """CONSISTENCY_* (tst_CUDAGenerator.h:226-511): repeated runs give the same grid;
    progress is monotone and ends at getMaxProgress()."""
    shape = cases.hexagon_shape()
    img, lib = cases.inputs(300, 300, 64, 40, Type.CIEDE2000)
    gen = generate(shape, img, lib, Type.CIEDE2000, rep=(3, 200))
    seen = []
    gen.connectProgress(seen.append)
    first = gen.getBestFits()[0]
    state = G.getGridState(gen.getCellGroup(), 300, 300)
    for _ in range(3):
        del seen[:]
        gen.setGridState(state)
        assert gen.generateBestFits()
        assert np.array_equal(gen.getBestFits()[0], first)
        assert seen == sorted(seen) and seen[-1] == gen.getMaxProgress()


def test_split_score_resolve_equals_generate(oracle, engine):
    """The cell-sharded form: two shards scored separately, candidate lists
    concatenated, one resolve -- same grid as mm_generate and as the oracle."""
    shape = cases.flippy_shape()
    group = cases.group_of(shape)
    t, rep = Type.CIEDE2000, (2, 250)
    img, lib = cases.inputs(150, 170, 24, 90, t)
    state = G.getGridState(group, 150, 170)[0]
    cells = G.cell_records(group, state, 150, 170)
    engine.set_library(lib); engine.set_masks(shape.masks4()); engine.set_target(img)
    whole, whole_score = engine.generate(cells, state.shape[1], state.shape[0], int(t), *rep)
    k = engine.candidates_per_cell(*rep)
    assert k == 2 * 5 + 2 + 2
    half = len(cells) // 2
    i0, s0 = engine.score_cells(cells[:half], int(t), k)
    i1, s1 = engine.score_cells(cells[half:], int(t), k)
    assert (np.diff(s0, axis=1) >= 0).all()
    best, score = engine.resolve(cells, state.shape[1], state.shape[0], int(t), rep[0], rep[1], k,
                                 np.concatenate([i0, i1]), np.concatenate([s0, s1]))
    assert np.array_equal(best, whole) and np.array_equal(score[state >= 0], whole_score[state >= 0])
    want, _, _ = oracle.generate_best_fits(shape_of(shape), img, lib, shape.masks4(), int(t), *rep)
    assert np.array_equal(np.where(best == len(lib), -1, best), want)


def test_error_behaviour_of_the_boundary():
    e = Engine(0)
    rec = np.zeros(1, dtype=G.CELL_DTYPE)
    rec["row_max"] = rec["col_max"] = 8
    with pytest.raises(EngineError) as err:                    # nothing staged
        e.generate(rec, 3, 3, 0)
    assert err.value.status == 5
    lib = np.zeros((3, 8, 8, 3), np.float32)
    e.set_library(lib); e.set_masks(np.full((4, 8, 8), 255, np.uint8)); e.set_target(np.zeros((8, 8, 3), np.float32))
    with pytest.raises(EngineError) as err:                    # ColourDifference::Type::MAX
        e.generate(rec, 3, 3, 3)
    assert err.value.status == 2
    bad = rec.copy(); bad["x"] = 4                             # reads past the image
    with pytest.raises(EngineError) as err:
        e.generate(bad, 3, 3, 0)
    assert err.value.status == 2 and "cell 0" in str(err.value)
    e.set_masks(np.full((4, 4, 4), 255, np.uint8))             # mask size != library size
    with pytest.raises(EngineError) as err:
        e.generate(rec, 3, 3, 0)
    assert err.value.status == 2
    with pytest.raises(EngineError):
        Engine(1 << 20)                                        # invalid device
    # all-identical library: index 0 wins everywhere (strict <)
    e.set_masks(np.full((4, 8, 8), 255, np.uint8))
    rec["grid_x"] = rec["grid_y"] = 1
    best, _ = e.generate(rec, 3, 3, 0)
    assert best[1, 1] == 0
    e.close()


def test_cancel_returns_false():
    shape = cases.square_shape(16)
    img, lib = cases.inputs(64, 64, 16, 10, Type.RGB_EUCLIDEAN)
    group = cases.group_of(shape)
    gen = HIPPhotomosaicGenerator(0)
    gen.setMainImage(img); gen.setLibrary(lib); gen.setCellGroup(group)
    gen.setGridState(G.getGridState(group, 64, 64))
    gen.cancel()
    assert gen.generateBestFits() is False


def test_empty_cell_list(engine):
    lib = np.zeros((3, 8, 8, 3), np.float32)
    engine.set_library(lib); engine.set_masks(np.full((4, 8, 8), 255, np.uint8))
    engine.set_target(np.zeros((8, 8, 3), np.float32))
    best, score = engine.generate(np.zeros(0, dtype=G.CELL_DTYPE), 4, 3, 0)
    assert (best == 3).all() and np.isnan(score).all()


def test_sharded_device_path_world1(oracle, engine):
    """ShardedBestFit with one rank: candidate lists written straight into torch CUDA
    tensors (mm_score_cells outputs_on_device), resolve reading device memory -- the
    multi-GPU step minus the collective (which the gloo test covers)."""
    import torch
    from mosaicmagnifique_amd.dist import ShardedBestFit
    shape = cases.hexagon_shape()
    group = cases.group_of(shape)
    t, rep = Type.CIEDE2000, (2, 250)
    img, lib = cases.inputs(260, 300, 64, 70, t)
    state = G.getGridState(group, 260, 300)[0]
    cells = G.cell_records(group, state, 260, 300)
    engine.set_library(lib); engine.set_masks(shape.masks4()); engine.set_target(img)
    whole, whole_score = engine.generate(cells, state.shape[1], state.shape[0], int(t), *rep)
    best, score = ShardedBestFit(engine, 0, 1).generate(cells, state.shape[1], state.shape[0], int(t), *rep)
    assert np.array_equal(best, whole) and np.array_equal(score[state >= 0], whole_score[state >= 0])
    assert torch.cuda.is_available()


@pytest.mark.parametrize("wave_resolve", list(RESOLVE_MODES))
def test_resolve_forms_on_ragged_rows(oracle, wave_resolve):
    """Hexagon rows are offset and ragged (x = -1 cells exist on some rows only) and the
    repeat window spans several rows: the row-pipelined resolve must wait on every row
    of the window, not just the one above."""
    shape = cases.hexagon_shape()
    img, lib = cases.inputs(500, 430, 64, 30, Type.RGB_EUCLIDEAN, lattice=500)
    compare(oracle, shape, img, lib, Type.RGB_EUCLIDEAN, rep=(3, 2000), wave_resolve=wave_resolve)
    compare(oracle, shape, img, lib, Type.RGB_EUCLIDEAN, rep=(1, 100000), wave_resolve=wave_resolve)


def test_many_small_batches_progress_and_cancel_between_batches(oracle, engine):
    """Cells are scored in device batches; with a 7-cell cap a 6 x 7 grid takes 6 batches.
    Same result, progress after every batch, and a cancel raised from the progress callback
    stops at the next batch boundary (PhotomosaicGeneratorBase::cancel, :217-220)."""
    import ctypes
    from mosaicmagnifique_amd.engine import Cancelled
    shape = cases.flippy_shape()
    group = cases.group_of(shape)
    t, rep = Type.CIEDE2000, (2, 120)
    img, lib = cases.inputs(120, 130, 24, 60, t)
    state = G.getGridState(group, 120, 130)[0]
    cells = G.cell_records(group, state, 120, 130)
    engine.set_library(lib); engine.set_masks(shape.masks4()); engine.set_target(img)
    whole, whole_score = engine.generate(cells, state.shape[1], state.shape[0], int(t), *rep)
    engine.set_option("max_batch", 7)
    try:
        seen = []
        best, score = engine.generate(cells, state.shape[1], state.shape[0], int(t), *rep, progress=seen.append)
        assert np.array_equal(best, whole) and np.array_equal(score[state >= 0], whole_score[state >= 0])
        assert seen == list(range(7, len(cells), 7)) + [len(cells)]
        want, _, _ = oracle.generate_best_fits(shape_of(shape), img, lib, shape.masks4(), int(t), *rep)
        assert np.array_equal(np.where(best == len(lib), -1, best), want)
        flag = ctypes.c_int(0)
        calls = []

        def stop_after_two(done):
            calls.append(done)
            if len(calls) == 2:
                flag.value = 1
        with pytest.raises(Cancelled):
            engine.generate(cells, state.shape[1], state.shape[0], int(t), *rep, progress=stop_after_two, cancel=flag)
        assert calls == [7, 14]
    finally:
        engine.set_option("max_batch", 0)


@pytest.mark.parametrize("wave_resolve", list(RESOLVE_MODES))
def test_negative_repeat_addition_promotes_any_candidate(oracle, wave_resolve):
    """setRepeat takes any int: a negative addition is a bonus for repeating neighbours, so
    the winner may be ANY library image and the whole score row is the candidate list."""
    shape = cases.square_shape(16)
    img, lib = cases.inputs(80, 96, 16, 90, Type.CIE76, lattice=96)
    gen = compare(oracle, shape, img, lib, Type.CIE76, rep=(2, -4000), wave_resolve=wave_resolve)
    assert gen.m_engine.candidates_per_cell(2, -4000) == 90
    fits = gen.getBestFits()[0]
    inner = fits[2:, 2:]
    assert (inner[:, 1:] == inner[:, :-1]).mean() > 0.5        # the bonus makes neighbours agree


@pytest.mark.parametrize("spec_k", [0, 2, 3, 4, 6, 10])
def test_speculative_short_lists_are_verified_and_deepened(oracle, engine, spec_k):
    """mm_generate scores short candidate lists first and lets the resolve verify each pick;
    unverified cells are deepened to the full list.  spec_k = 2 / 3 make that happen all the
    time, 0 switches speculation off: the grid never changes."""
    shape = cases.square_shape(16)
    group = cases.group_of(shape)
    t, rep = Type.CIE76, (3, 900)
    img, lib = cases.inputs(160, 176, 16, 120, t, lattice=200)       # smooth target: neighbours collide
    state = G.getGridState(group, 160, 176)[0]
    cells = G.cell_records(group, state, 160, 176)
    engine.set_library(lib); engine.set_masks(shape.masks4()); engine.set_target(img)
    want, want_score, _ = oracle.generate_best_fits(shape_of(shape), img, lib, shape.masks4(), int(t), *rep)
    engine.set_option("spec_k", spec_k)
    try:
        best, score = engine.generate(cells, state.shape[1], state.shape[0], int(t), *rep)
        tm = engine.timing()
    finally:
        engine.set_option("spec_k", 4)
    assert np.array_equal(np.where(best == len(lib), -1, best), want)
    valid = state >= 0
    assert (np.abs(score[valid] - want_score[valid]) <= 1e-12 * want_score[valid]).all()
    full = engine.candidates_per_cell(*rep)
    assert full == 3 * 7 + 3 + 2
    if spec_k == 0:
        assert tm["spec_rounds"] == 1 and tm["deepened_cells"] == 0
    elif spec_k <= 3:
        assert tm["spec_rounds"] >= 2 and tm["deepened_cells"] > 0


def test_split_api_speculation_matches_generate(oracle, engine):
    """The speculate / verify / deepen loop driven from the host through mm_score_cells_partial
    and mm_resolve_verify (what the multi-GPU step does), on two shards."""
    shape = cases.flippy_shape()
    group = cases.group_of(shape)
    t, rep = Type.CIEDE2000, (3, 400)
    img, lib = cases.inputs(150, 170, 24, 90, t, lattice=300)
    state = G.getGridState(group, 150, 170)[0]
    cells = G.cell_records(group, state, 150, 170)
    engine.set_library(lib); engine.set_masks(shape.masks4()); engine.set_target(img)
    whole, whole_score = engine.generate(cells, state.shape[1], state.shape[0], int(t), *rep)
    k = engine.candidates_per_cell(*rep)
    ks = engine.speculative_candidates(*rep, spec_k=3)
    assert ks == 3 < k
    half = len(cells) // 2
    i0, s0 = engine.score_cells(cells[:half], int(t), k, k_fill=ks)
    i1, s1 = engine.score_cells(cells[half:], int(t), k, k_fill=ks)
    idx, sc = np.concatenate([i0, i1]), np.concatenate([s0, s1])
    assert (idx[:, ks:] == -1).all() and (idx[:, :ks] != -1).all()
    deep = np.zeros(len(cells), np.uint8)
    rounds = 0
    while True:
        best, score, insufficient = engine.resolve(cells, state.shape[1], state.shape[0], int(t), rep[0], rep[1], k,
                                                   idx, sc, deep=deep, k_fill=ks)
        rounds += 1
        todo = np.nonzero((insufficient != 0) & (deep == 0))[0]
        if len(todo) == 0:
            break
        di, ds = engine.score_cells(cells[todo], int(t), k)
        idx[todo], sc[todo] = di, ds
        deep[todo] = 1
    assert rounds >= 2 and deep.sum() > 0
    assert np.array_equal(best, whole) and np.array_equal(score[state >= 0], whole_score[state >= 0])


@pytest.mark.parametrize("t", [Type.RGB_EUCLIDEAN, Type.CIE76, Type.CIEDE2000])
@pytest.mark.parametrize("rep", [(0, 0), (3, 700)], ids=["norepeat", "repeats"])
def test_consistency_over_iterations(t, rep):
    """CUDAGeneratorFixture CONSISTENCY_* (test/tst_CUDAGenerator.h:62-85, :226-330): the same
    generator run ITERATIONS times from a fresh grid state returns the same best fits every time
    (atomics, dynamic work pulls and the speculative path must not leak run-to-run variation)."""
    shape = cases.hexagon_shape()
    group = cases.group_of(shape)
    img, lib = cases.inputs(300, 410, 64, 90, t, lattice=96)
    gen = HIPPhotomosaicGenerator(0)
    gen.setMainImage(img); gen.setLibrary(lib); gen.setColourDifference(t); gen.setCellGroup(group)
    gen.setRepeat(*rep)
    last = None
    for _ in range(5):
        gen.setGridState(G.getGridState(group, img.shape[0], img.shape[1]))
        assert gen.generateBestFits(), gen.lastError
        fits, scores = gen.getBestFits()[0], gen.getBestScores()[0]
        if last is not None:
            assert np.array_equal(fits, last[0])
            assert np.array_equal(scores, last[1], equal_nan=True)         # bit-identical, not just close
        last = (fits, scores)


@pytest.mark.gpu
@pytest.mark.parametrize("t", [Type.RGB_EUCLIDEAN, Type.CIEDE2000])
@pytest.mark.parametrize("size,n_lib", [(1, 1), (1, 7), (2, 2), (3, 3), (5, 1), (2, 40)])
def test_smallest_cells_and_libraries(oracle, t, size, n_lib):
    """One-pixel cells, one-image libraries: every list is longer than the library, every plane is one padded group."""
    shape = cases.square_shape(size)
    img, lib = cases.inputs(7 * size + 1, 9 * size, size, n_lib, t, lattice=4)
    for rep in ((0, 0), (1, 10), (3, 1000)):
        compare(oracle, shape, img, lib, t, rep)


@pytest.mark.gpu
@pytest.mark.parametrize("use_filter", [1, 0], ids=["two_tier", "fp64_only"])
def test_repeat_window_longer_than_the_seed_list(oracle, use_filter):
    """Repeat range 12: a cell's window holds up to 312 earlier cells, more candidates than the 256 seeds the early exit
    can take its limit from -- the filter runs without limits; the penalty is small against the score spread, so the
    window actually pushes picks around."""
    t, size, n_lib = Type.CIEDE2000, 16, 400
    shape = cases.square_shape(size)
    img, lib = cases.inputs(20 * size, 20 * size, size, n_lib, t, lattice=64)
    gen = compare(oracle, shape, img, lib, t, (12, 60), use_filter)
    assert gen.m_engine.candidates_per_cell(12, 60) > 256


@pytest.mark.gpu
@pytest.mark.parametrize("t", [Type.RGB_EUCLIDEAN, Type.CIEDE2000])
@pytest.mark.parametrize("kind", ["empty", "one_pixel", "one_row", "flip_sensitive_corner"])
def test_masks_with_next_to_nothing_in_them(oracle, t, kind):
    """A mask with no active pixel at all (every candidate scores the bare repeat penalty: the lowest index, or the
    least penalised, wins), one active pixel, one row, and a single corner pixel under flips (four different masks)."""
    from mosaicmagnifique_amd.cellshape import CellShape
    size = 24
    mask = np.zeros((size, size), np.uint8)
    if kind == "one_pixel":
        mask[size // 2, size // 3] = 255
    elif kind == "one_row":
        mask[5, :] = 255
    elif kind == "flip_sensitive_corner":
        mask[0, 0] = 255
    shape = CellShape(mask)
    if kind == "flip_sensitive_corner":
        shape.alternateColFlipHorizontal = True
        shape.alternateRowFlipVertical = True
    img, lib = cases.inputs(6 * size + 5, 7 * size, size, 60, t, lattice=16)
    for rep in ((0, 0), (2, 25)):
        for use_filter in (1, 0):
            compare(oracle, shape, img, lib, t, rep, use_filter)


@pytest.mark.gpu
@pytest.mark.parametrize("t", [Type.RGB_EUCLIDEAN, Type.CIE76, Type.CIEDE2000])
@pytest.mark.parametrize("size", [16, 64])
def test_flat_colour_content(oracle, t, size):
    """Exactly constant colours on both sides -- black, white, greys, the saturated primaries, a blue at the hue where
    R_T peaks, near-opposite hues -- and every library colour twice: scores that are exactly zero, exact ties between
    duplicates (lowest index wins), blocks of zero chroma (open hue arcs) and of zero-width arcs in the bound."""
    colours = np.array([[0, 0, 0], [255, 255, 255], [128, 128, 128], [127, 127, 127], [0, 0, 255], [0, 255, 0],
                        [255, 0, 0], [255, 0, 40], [0, 255, 255], [255, 255, 0], [255, 0, 255], [1, 0, 0], [254, 255, 255],
                        [200, 30, 30], [30, 30, 200], [60, 200, 60], [20, 20, 20], [235, 235, 235]], np.uint8)      # BGR
    n = len(colours)
    copies = 6                                                      # exact duplicates: ties at every score; 108 images,
    lib_u8 = np.empty((copies * n, size, size, 3), np.uint8)        # so the short lists are shorter than the library
    for i in range(copies * n):                                     # and the two-tier path with its early exit runs
        lib_u8[i] = colours[i % n]
    rows_c, cols_c = 9, 11
    img_u8 = np.empty((rows_c * size, cols_c * size, 3), np.uint8)
    rng = np.random.default_rng(12)
    for gy in range(rows_c):
        for gx in range(cols_c):
            c = colours[rng.integers(n)].astype(int)
            if (gy + gx) % 3 == 0:                                  # a colour nothing in the library matches exactly
                c = np.clip(c + rng.integers(-3, 4, 3), 0, 255)
            img_u8[gy * size:(gy + 1) * size, gx * size:(gx + 1) * size] = c
    from mosaicmagnifique_amd import synth
    img, lib = synth.preprocess(img_u8, t), synth.preprocess(lib_u8, t)
    shape = cases.square_shape(size)
    for rep in ((0, 0), (1, 5), (3, 100000)):
        for use_filter in (1, 0):
            gen = compare(oracle, shape, img, lib, t, rep, use_filter)
            if use_filter and rep == (0, 0):
                assert gen.timing()["rescore_ms"] > 0           # the two-tier path did run


@pytest.mark.gpu
@pytest.mark.parametrize("size", [48, 64])
def test_smooth_content(oracle, size):
    """Noise-free gradients on both sides (what photographs look like next to the synthetic library's +-32 of noise):
    block statistics are tight, hue arcs narrow, the bound drops nearly every candidate in stage 1 and the survivors
    are near-ties."""
    t, n_lib = Type.CIEDE2000, 160
    rng = np.random.default_rng(77)
    yy, xx = np.mgrid[0:size, 0:size].astype(np.float64) / size
    lib_u8 = np.empty((n_lib, size, size, 3), np.uint8)
    for i in range(n_lib):
        base = rng.uniform(20, 235, 3)
        gy, gx = rng.uniform(-40, 40, 3), rng.uniform(-40, 40, 3)
        lib_u8[i] = np.clip(base + yy[..., None] * gy + xx[..., None] * gx, 0, 255).round().astype(np.uint8)
    rows, cols = 5 * size + size // 2, 6 * size
    Y, X = np.mgrid[0:rows, 0:cols].astype(np.float64)
    img_u8 = np.stack([127 + 100 * np.sin(X / 210 + c) * np.cos(Y / 170 - c) for c in (0.0, 1.1, 2.3)], -1)
    img_u8 = np.clip(img_u8, 0, 255).round().astype(np.uint8)
    from mosaicmagnifique_amd import synth
    img, lib = synth.preprocess(img_u8, t), synth.preprocess(lib_u8, t)
    shape = cases.square_shape(size)
    for rep in ((0, 0), (2, 200)):
        gen = compare(oracle, shape, img, lib, t, rep)
        tm = gen.timing()
        assert tm["band_alarms"] == 0
    assert tm["bound_dropped"] > 0


def test_timing_log_uses_the_reference_s_phase_ids(oracle):
    """timingLog(): the last run's times under the ids the reference's TimingLogger records on its device path
    (CUDAPhotomosaicGenerator.cpp:45-358,376-506) -- every id present, every time non-negative, the parts within the whole."""
    shape = cases.square_shape(32)
    img, lib = cases.inputs(160, 192, 32, 90, Type.CIEDE2000)
    gen = compare(oracle, shape, img, lib, Type.CIEDE2000, (2, 300))
    log = gen.timingLog()
    assert set(log) == {"generateBestFits", "Preprocess", "cudaMalloc", "cudaMemcpy", "DiffReduce", "Flatten", "Repeats+FindLowest"}
    assert all(v >= 0 for v in log.values()) and log["generateBestFits"] > 0 and log["DiffReduce"] > 0
    assert log["cudaMemcpy"] + log["DiffReduce"] + log["Flatten"] + log["Repeats+FindLowest"] <= log["generateBestFits"] * 1.05
