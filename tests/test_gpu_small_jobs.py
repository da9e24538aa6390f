"""The reference's everyday regime -- about a thousand cells against a couple of hundred images
(Benchmark/Benchmark_Generator.h:125-183, MainWindow.cpp:558-638) -- on the engine's OWN defaults: the whole
generate is one submission read back in one result block (mm_timing.fast_path / host_syncs), jobs below the work
threshold run without seeds, early exit and phase events, and the grid still is the oracle's, cell by cell.
(The rest of the suite turns the thresholds off so that small scenarios exercise the seeds and the bound.)"""
import os

import numpy as np
import pytest

import cases
from mosaicmagnifique_amd import GridUtility as G
from mosaicmagnifique_amd.colourdifference import Type
from mosaicmagnifique_amd.engine import Engine
from oracle_ffi import shape_of

pytestmark = pytest.mark.gpu


@pytest.fixture()
def default_engine():
    e = Engine(0)
    e.set_option("phase_timing", -1)           # the library's defaults (conftest turns the thresholds off for the suite)
    e.set_option("prune_min_kpairs", -1)
    yield e
    e.close()


def threads():
    return max(1, min(16, len(os.sched_getaffinity(0))))


def check_against_oracle(oracle, shape, img, lib, t, rep, cells, best, score, n_lib):
    grid = np.where(best < n_lib, best, -1)
    pos = np.stack([cells["grid_y"], cells["grid_x"]], axis=1)
    picks, scores = oracle.picks_given_grid(shape_of(shape), img, lib, shape.masks4(), int(t), rep[0], rep[1], grid, pos,
                                            threads=threads())
    got = best[pos[:, 0], pos[:, 1]]
    assert np.array_equal(np.where(got == n_lib, -1, got), picks), np.flatnonzero(np.where(got == n_lib, -1, got) != picks)[:5]
    ok = picks >= 0
    tol = 1e-4 if t == Type.CIEDE2000 else 1e-12
    assert (np.abs(score[pos[ok, 0], pos[ok, 1]] - scores[ok]) <= tol * np.abs(scores[ok]) + 1e-300).all()


def test_config_a_is_one_submission(oracle, default_engine):
    """Config A (512 x 512, 32 x 32 cells, 100 images, RGB-Euclidean): 26 M pixel pairs -- below the threshold, so no
    seeds, no early exit, no phase events; one host wait; every cell equals the oracle's sequential run."""
    e = default_engine
    t, L, size = Type.RGB_EUCLIDEAN, 100, 32
    shape = cases.square_shape(size)
    img, lib = cases.inputs(512, 512, size, L, t, lattice=64)
    group = cases.group_of(shape)
    state = G.getGridState(group, 512, 512)[0]
    cells = G.cell_records(group, state, 512, 512)
    e.set_library(lib); e.set_masks(shape.masks4()); e.set_target(img)
    e.generate(cells, state.shape[1], state.shape[0], int(t))       # (the first call also builds the library planes)
    best, score = e.generate(cells, state.shape[1], state.shape[0], int(t))
    tm = e.timing()
    assert tm["fast_path"] == 1 and tm["host_syncs"] == 1
    assert tm["seed_ms"] == 0 and tm["score_ms"] == 0 and tm["bound_dropped"] == 0      # unpruned, no events
    want, want_score, n = oracle.generate_best_fits(shape_of(shape), img, lib, shape.masks4(), int(t), 0, 0)
    assert n == 256 and np.array_equal(np.where(best < L, best, -1), want)
    valid = want >= 0
    assert (np.abs(score[valid] - want_score[valid]) <= 1e-12 * want_score[valid]).all()     # parallel vs sequential fp64 sum


@pytest.mark.parametrize("t,rep", [(Type.CIEDE2000, (0, 0)), (Type.CIEDE2000, (3, 400)), (Type.CIE76, (2, 150)),
                                   (Type.RGB_EUCLIDEAN, (4, 50))])
def test_everyday_job_on_the_defaults(oracle, default_engine, t, rep):
    """~1000 cells x 200 images of 32 x 32 (the GUI's default detail of a 64 px cell): with a repeat penalty the
    short speculative lists leave some picks unverified, which the result block reports -- the call then goes straight
    to the deepening step, without a second resolve of what it already knows."""
    e = default_engine
    L, size = 200, 32
    shape = cases.square_shape(size)
    img, lib = cases.inputs(32 * 32, 32 * 32, size, L, t, lattice=48)
    group = cases.group_of(shape)
    state = G.getGridState(group, img.shape[0], img.shape[1])[0]
    cells = G.cell_records(group, state, img.shape[0], img.shape[1])
    assert len(cells) == 1024
    e.set_library(lib); e.set_masks(shape.masks4()); e.set_target(img)
    e.generate(cells, state.shape[1], state.shape[0], int(t), *rep)
    best, score = e.generate(cells, state.shape[1], state.shape[0], int(t), *rep)
    tm = e.timing()
    assert tm["band_alarms"] == 0
    if rep[0] == 0:
        assert tm["fast_path"] == 1 and tm["host_syncs"] == 1
    else:
        assert tm["fast_path"] == 1 or tm["deepened_cells"] > 0      # unverified picks are the one reason to go on
    check_against_oracle(oracle, shape, img, lib, t, rep, cells, best, score, L)
    # the step-by-step path gives the same grid and the same score bits
    e.set_option("fast_path", 0)
    try:
        slow_best, slow_score = e.generate(cells, state.shape[1], state.shape[0], int(t), *rep)
    finally:
        e.set_option("fast_path", 1)
    assert e.timing()["fast_path"] == 0
    assert np.array_equal(slow_best, best) and np.array_equal(slow_score[state >= 0], score[state >= 0])


def test_cells_without_an_active_pixel_stay_on_the_device(oracle, default_engine):
    """Offset shapes leave slivers whose bound misses the mask altogether (config C has 22): such a cell scores exactly
    0 against every image.  Its list is written on the device (the K lowest indices, score 0) instead of sending the
    cell through the fp64-only path, and the pick is the reference's: the lowest index the repeat window allows."""
    e = default_engine
    t, L = Type.CIEDE2000, 120
    shape = cases.hexagon_shape()
    img, lib = cases.inputs(400, 500, 64, L, t)
    group = cases.group_of(shape)
    state = G.getGridState(group, img.shape[0], img.shape[1])[0]
    cells = G.cell_records(group, state, img.shape[0], img.shape[1])
    masks = shape.masks4() != 0
    empty = [i for i, c in enumerate(cells)
             if not masks[c["mask_id"]][c["row_min"]:c["row_max"], c["col_min"]:c["col_max"]].any()]
    assert empty, "the scenario is meant to hold cells without an active pixel"
    e.set_library(lib); e.set_masks(shape.masks4()); e.set_target(img)
    for rep in ((0, 0), (2, 250)):
        best, score = e.generate(cells, state.shape[1], state.shape[0], int(t), *rep)
        tm = e.timing()
        assert tm["dense_cells"] == 0 and tm["dense_rows"] == 0
        for i in empty:                      # base score 0: the total is a whole number of penalties
            total = score[cells[i]["grid_y"], cells[i]["grid_x"]]
            assert total == 0 if rep[1] == 0 else total % rep[1] == 0
        check_against_oracle(oracle, shape, img, lib, t, rep, cells, best, score, L)


@pytest.mark.parametrize("size,t,rep", [(8, Type.CIEDE2000, (2, 300)), (8, Type.RGB_EUCLIDEAN, (0, 0)), (6, Type.CIE76, (3, 40)),
                                        (3, Type.CIEDE2000, (0, 0)), (1, Type.CIE76, (1, 25)), (5, Type.RGB_EUCLIDEAN, (4, -30))])
def test_tiny_detail_cells_one_wave_per_cell(oracle, default_engine, size, t, rep):
    """Detail cells of <= 64 pixels (the reference's harness goes down to cell size 8 at detail 20 %: ONE pixel,
    Benchmark/Benchmark_Generator.h:163-171) take the one-wave-per-cell path: every cell against the oracle, the general
    path (option "tiny" 0) gives the same grid and the same score bits up to the summation order; a negative addition
    asks for the whole row (K = L)."""
    e = default_engine
    L = 150
    yy, xx = np.mgrid[0:size, 0:size]
    mask = np.where((xx + yy) % 5 != 4, 255, 0).astype(np.uint8) if size >= 5 else np.full((size, size), 255, np.uint8)
    from mosaicmagnifique_amd.cellshape import CellShape
    shape = CellShape(mask)
    if size >= 5:
        shape.alternateRowOffset = 2                       # offset rows: clipped slivers at the borders
        shape.alternateColFlipHorizontal = True
    rows, cols = 37 * max(size, 2) + 3, 41 * max(size, 2) + 1
    img, lib = cases.inputs(rows, cols, size, L, t, lattice=max(4 * size, 8))
    group = cases.group_of(shape)
    state = G.getGridState(group, rows, cols)[0]
    cells = G.cell_records(group, state, rows, cols)
    assert len(cells) > 1000
    e.set_library(lib); e.set_masks(shape.masks4()); e.set_target(img)
    e.generate(cells, state.shape[1], state.shape[0], int(t), *rep)
    best, score = e.generate(cells, state.shape[1], state.shape[0], int(t), *rep)
    tm = e.timing()
    assert tm["rescore_ms"] == 0 and tm["seed_ms"] == 0           # neither filter tier nor seeds ran
    check_against_oracle(oracle, shape, img, lib, t, rep, cells, best, score, L)
    e.set_option("tiny", 0)
    try:
        gbest, gscore = e.generate(cells, state.shape[1], state.shape[0], int(t), *rep)
    finally:
        e.set_option("tiny", 1)
    valid = state >= 0
    assert np.array_equal(gbest, best)
    assert (np.abs(gscore[valid] - score[valid]) <= 1e-12 * np.abs(score[valid]) + 1e-300).all()


@pytest.mark.parametrize("t,rep", [(Type.RGB_EUCLIDEAN, (0, 0)), (Type.CIE76, (3, 120))])
def test_euclid_filter_stage_1_one_candidate_per_lane(oracle, engine, t, rep):
    """The RGB / CIE76 filter judges the block bound one candidate per lane before it spends a wave on a candidate (on
    benchmark-like content the bound alone drops nearly all of them): hexagon cells (clipped border cells use the bound on
    their whole blocks), 700 images, every cell against the oracle; option euclid_stage1 = 0 gives the same grid and bits."""
    L = 700
    shape = cases.hexagon_shape()
    img, lib = cases.inputs(900, 1100, 64, L, t, lattice=96)
    group = cases.group_of(shape)
    state = G.getGridState(group, img.shape[0], img.shape[1])[0]
    cells = G.cell_records(group, state, img.shape[0], img.shape[1])
    engine.set_library(lib); engine.set_masks(shape.masks4()); engine.set_target(img)
    best, score = engine.generate(cells, state.shape[1], state.shape[0], int(t), *rep)
    tm = engine.timing()
    assert tm["bound_dropped"] > 0.5 * len(cells) * L and tm["visited_groups"] > 0 and tm["band_alarms"] == 0
    check_against_oracle(oracle, shape, img, lib, t, rep, cells, best, score, L)
    engine.set_option("euclid_stage1", 0)
    try:
        b2, s2 = engine.generate(cells, state.shape[1], state.shape[0], int(t), *rep)
    finally:
        engine.set_option("euclid_stage1", 1)
    valid = state >= 0
    assert np.array_equal(b2, best) and np.array_equal(s2[valid], score[valid])
