"""BASELINE.json's full-size configurations on the GPU (the shape of the reference's
whole-grid CPU == device pins, test/tst_CUDAGenerator.h:163-212,530-815):

* config A: the whole grid against the oracle;
* config B at 1024 x 1024 (same cells, same library): the whole grid against the oracle;
* config D: the first TWO FULL ROWS (128 cells, so the repeat window's rows-above half is
  exercised) against the oracle, verified cell by cell in parallel on the host
  cores (oracle_ffi.picks_given_grid: the induction over the raster order); 32 random
  interior cells' two lowest base scores against the oracle without early exit;
* configs C, D and E's one-rank shard: the default path (early-exit bound + fp32 filter +
  fp64 arbiter) gives the same whole grid and bit-equal scores as the unpruned path and as
  the fp64-only path -- the bound and the filter band are content dependent, so they are
  checked at the sizes that are benchmarked, over every cell;
* the older properties: oracle parity on a raster-order prefix, library-reversal
  equivariance, shard/whole agreement."""
import os

import numpy as np
import pytest

import cases
from mosaicmagnifique_amd import GridUtility as G
from mosaicmagnifique_amd import synth
from mosaicmagnifique_amd.colourdifference import Type
from oracle_ffi import shape_of

pytestmark = pytest.mark.gpu


def stage(engine, shape, img, lib):
    engine.set_library(lib)
    engine.set_masks(shape.masks4())
    engine.set_target(img)


def prefix_parity(oracle, shape, img, lib, t, rep, best, score, n_prefix, rel_tol):
    want, want_score, done = oracle.generate_best_fits(
        shape_of(shape), img, lib, shape.masks4(), int(t), rep[0], rep[1], max_cells=n_prefix)
    recs = G.cell_records(cases.group_of(shape), oracle.grid_state(shape_of(shape), img.shape[1], img.shape[0]),
                          img.shape[0], img.shape[1])[:done]
    for r in recs:
        gy, gx = r["grid_y"], r["grid_x"]
        assert best[gy, gx] == want[gy, gx], (gy, gx)
        assert abs(score[gy, gx] - want_score[gy, gx]) <= rel_tol * want_score[gy, gx]


_e_shard = {}


def e_shard_inputs():
    """Config E as ONE rank of 8 sees it: the first ceil(29 241 / 8) cells, the target band
    holding them (full width 8192), the 10 000-image library."""
    if not _e_shard:
        t, L, size = Type.CIEDE2000, 10000, 48
        shape = cases.square_shape(size)
        group = cases.group_of(shape)
        full_state = G.getGridState(group, 8192, 8192)[0]
        assert int((full_state >= 0).sum()) == 171 * 171
        n_shard = -(-171 * 171 // 8)
        band_rows = (n_shard // 171 + 2) * size
        img = synth.preprocess(synth.target_u8(band_rows, 8192, lattice=64), t)
        lib = synth.preprocess(synth.library_u8(L, size), t)
        cells = G.cell_records(group, full_state, 8192, 8192)[:n_shard]
        _e_shard["v"] = (t, L, shape, img, lib, cells, n_shard)
    return _e_shard["v"]


def host_threads(cap=16):
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    return max(1, min(n, cap))


def with_options(engine, options, fn):
    """fn() with the engine options set, then the defaults restored (session-wide engine)."""
    try:
        for k, v in options.items():
            engine.set_option(k, v)
        return fn()
    finally:
        for k in options:
            engine.set_option(k, 1)


def assert_same_on_every_path(engine, cells, gw, gh, t, rep, valid, L):
    """Default path == early exit off == fp64 only: whole grid, bit-equal scores."""
    run = lambda: engine.generate(cells, gw, gh, int(t), *rep)
    best, score = run()
    tm = engine.timing()
    assert (best[valid] < L).all() and (best[~valid] == L).all()
    assert tm["seed_ms"] > 0 and tm["rescore_ms"] > 0          # the pruned two-tier path did run
    nb, ns = with_options(engine, {"prune": 0}, run)
    assert np.array_equal(nb, best), int((nb != best).sum())
    assert np.array_equal(ns[valid], score[valid])
    for forced in (1, 0):                                      # the packed-fp16 pre-tier forced on and off (default: by the size of the job)
        try:
            engine.set_option("pretier", forced)
            pb, ps = run()
        finally:
            engine.set_option("pretier", -1)
        assert np.array_equal(pb, best), (forced, int((pb != best).sum()))
        assert np.array_equal(ps[valid], score[valid]), forced
    fb, fs = with_options(engine, {"filter": 0}, run)
    assert engine.timing()["rescore_ms"] == 0
    assert np.array_equal(fb, best), int((fb != best).sum())
    assert np.array_equal(fs[valid], score[valid])
    return best, score, tm


def test_config_a_rgb_512_square32_l100_whole_grid(oracle, engine):
    """Config A (BASELINE configs[0], the reference's own CPU-runnable case): 512 x 512 target,
    32 x 32 square cells, 100 images, RGB-Euclidean -- all 256 cells against the oracle's
    sequential run; with and without the repeat penalty."""
    t, L, size = Type.RGB_EUCLIDEAN, 100, 32
    shape = cases.square_shape(size)
    img, lib = cases.inputs(512, 512, size, L, t, lattice=64)
    group = cases.group_of(shape)
    state = G.getGridState(group, 512, 512)[0]
    cells = G.cell_records(group, state, 512, 512)
    assert len(cells) == 256
    stage(engine, shape, img, lib)
    valid = state >= 0
    for rep in ((0, 0), (4, 500)):
        best, score = engine.generate(cells, state.shape[1], state.shape[0], int(t), *rep)
        want, want_score, done = oracle.generate_best_fits(shape_of(shape), img, lib, shape.masks4(), int(t), *rep)
        assert done == 256
        assert np.array_equal(best[valid], want[valid]) and (best[~valid] == L).all()
        # Euclid terms are bit-identical; only the summation order differs (parallel vs sequential fp64)
        assert (np.abs(score[valid] - want_score[valid]) <= 1e-12 * want_score[valid]).all()


def test_config_b_reduced_rgb_1024_square64_l1000_whole_grid(oracle, engine):
    """Config B on a 1024 x 1024 target (256 of its 64 x 64 cells, the full 1000-image
    library): the whole grid against the oracle."""
    t, L, size = Type.RGB_EUCLIDEAN, 1000, 64
    shape = cases.square_shape(size)
    img, lib = cases.inputs(1024, 1024, size, L, t, lattice=64)
    group = cases.group_of(shape)
    state = G.getGridState(group, 1024, 1024)[0]
    cells = G.cell_records(group, state, 1024, 1024)
    assert len(cells) == 256
    stage(engine, shape, img, lib)
    best, score = engine.generate(cells, state.shape[1], state.shape[0], int(t))
    grid = np.where(best < L, best, -1)
    pos = np.stack([cells["grid_y"], cells["grid_x"]], axis=1)
    picks, scores = oracle.picks_given_grid(shape_of(shape), img, lib, shape.masks4(), int(t), 0, 0, grid, pos,
                                            threads=host_threads())
    assert np.array_equal(picks, best[pos[:, 0], pos[:, 1]])
    assert (np.abs(scores - score[pos[:, 0], pos[:, 1]]) <= 1e-12 * scores).all()


def test_config_b_rgb_2048_square64_l1000(oracle, engine):
    """Config B.  Properties: (1) first 48 cells == oracle; (2) reversing the library
    maps every chosen index i -> L-1-i with bit-identical scores (no ties in the
    synthetic library), i.e. the argmin does not depend on candidate position."""
    t, L, size = Type.RGB_EUCLIDEAN, 1000, 64
    shape = cases.square_shape(size)
    img, lib = cases.inputs(2048, 2048, size, L, t, lattice=64)
    group = cases.group_of(shape)
    state = G.getGridState(group, 2048, 2048)[0]
    cells = G.cell_records(group, state, 2048, 2048)
    assert len(cells) == 1024
    stage(engine, shape, img, lib)
    best, score = engine.generate(cells, state.shape[1], state.shape[0], int(t))
    valid = state >= 0
    assert (best[valid] < L).all() and (best[~valid] == L).all()
    prefix_parity(oracle, shape, img, lib, t, (0, 0), best, score, 48, 1e-12)

    engine.set_library(lib[::-1])
    rbest, rscore = engine.generate(cells, state.shape[1], state.shape[0], int(t))
    assert np.array_equal(rbest[valid], L - 1 - best[valid])
    assert np.array_equal(rscore[valid], score[valid])


def test_config_c_ciede2000_2048_hexagon_l1000(oracle, engine):
    """Config C: Hexagon.mcs mask, clipped border cells, CIEDE2000."""
    t, L = Type.CIEDE2000, 1000
    shape = cases.hexagon_shape()
    img, lib = cases.inputs(2048, 2048, 64, L, t, lattice=64)
    group = cases.group_of(shape)
    state = G.getGridState(group, 2048, 2048)[0]
    cells = G.cell_records(group, state, 2048, 2048)
    stage(engine, shape, img, lib)
    best, score = engine.generate(cells, state.shape[1], state.shape[0], int(t))
    valid = state >= 0
    assert (best[valid] < L).all()
    prefix_parity(oracle, shape, img, lib, t, (0, 0), best, score, 12, 1e-4)
    # clipped border cells whose visible part misses the mask have NO active pixel: every
    # candidate scores exactly 0 and index 0 wins without any reference-order re-sum
    assert engine.timing()["exact_rescored"] == 0
    # interior (unclipped) and clipped cells both present in the prefix? check one clipped cell explicitly
    clipped = cells[(cells["row_max"] - cells["row_min"] < 64) | (cells["col_max"] - cells["col_min"] < 64)]
    assert len(clipped) > 100
    # ... and at the benchmarked size against the ORACLE, not only against the engine's other paths: 64 unclipped and
    # 16 clipped cells drawn at random from the whole grid (the sample bench.py --config C reports parity on)
    random_cells_against_the_oracle(oracle, engine, shape, img, lib, t, cells, best, score, L)


def random_cells_against_the_oracle(oracle, engine, shape, img, lib, t, cells, best, score, L, rel_tol=1e-4):
    import bench
    sample, _ = bench.parity_sample(dict(rep=(0, 0), size=shape.getSize(), shape=shape), cells)
    assert len(sample) >= 64
    pos = np.stack([sample["grid_y"], sample["grid_x"]], axis=1)
    grid = np.where(best < L, best, -1)
    picks, scores = oracle.picks_given_grid(shape_of(shape), img, lib, shape.masks4(), int(t), 0, 0, grid, pos, threads=host_threads())
    got = best[pos[:, 0], pos[:, 1]]
    assert np.array_equal(np.where(got == L, -1, got), picks), np.flatnonzero(np.where(got == L, -1, got) != picks)
    ok = picks >= 0
    assert (np.abs(score[pos[ok, 0], pos[ok, 1]] - scores[ok]) <= rel_tol * np.abs(scores[ok]) + 1e-300).all()


def test_config_c_whole_grid_same_on_every_path(engine):
    """Config C (Hexagon mask, clipped border cells), all cells: default == unpruned == fp64-only."""
    t, L = Type.CIEDE2000, 1000
    shape = cases.hexagon_shape()
    img, lib = cases.inputs(2048, 2048, 64, L, t, lattice=64)
    group = cases.group_of(shape)
    state = G.getGridState(group, 2048, 2048)[0]
    cells = G.cell_records(group, state, 2048, 2048)
    stage(engine, shape, img, lib)
    assert_same_on_every_path(engine, cells, state.shape[1], state.shape[0], t, (0, 0), state >= 0, L)


def test_config_d_ciede2000_4096_l5000_repeat4(oracle, engine):
    """Config D (the headline): 4096 cells x 5000 images, repeat range 4 (+500).
    Oracle parity on the first 6 cells of the raster (they already see each other
    through the repeat window); every pick is a valid index; with a penalty this
    large no cell may repeat a neighbour that an unpenalised candidate beats --
    checked as: picks differ from the left neighbour far more often than without
    repeats."""
    t, L, size, rep = Type.CIEDE2000, 5000, 64, (4, 500)
    shape = cases.square_shape(size)
    img, lib = cases.inputs(4096, 4096, size, L, t, lattice=64)
    group = cases.group_of(shape)
    state = G.getGridState(group, 4096, 4096)[0]
    cells = G.cell_records(group, state, 4096, 4096)
    assert len(cells) == 4096
    stage(engine, shape, img, lib)
    best, score = engine.generate(cells, state.shape[1], state.shape[0], int(t), *rep)
    valid = state >= 0
    assert (best[valid] < L).all()
    prefix_parity(oracle, shape, img, lib, t, rep, best, score, 6, 1e-4)
    plain, _ = engine.generate(cells, state.shape[1], state.shape[0], int(t))
    inner = best[2:, 2:]
    same_rep = (inner[:, 1:] == inner[:, :-1]).mean()
    same_plain = (plain[2:, 2:][:, 1:] == plain[2:, 2:][:, :-1]).mean()
    assert same_rep <= same_plain


def test_config_d_first_two_rows_and_interior_cells_vs_oracle(oracle, engine):
    """Config D against the oracle where SURVEY 8(d) asks for it: the first 2 full rows
    (128 cells; from the second row on a cell's repeat window holds rows above it), each
    cell evaluated by the oracle exactly as the reference's raster loop reaches it with
    the earlier cells at the engine's picks -- equality on the whole prefix is equality
    with the sequential run (induction).  Then 32 random interior cells: their two lowest
    BASE scores (no penalty) from mm_score_cells against the oracle without early exit;
    the runner-up is the oracle's pick once the winner carries a large repeat value."""
    t, L, size, rep = Type.CIEDE2000, 5000, 64, (4, 500)
    shape = cases.square_shape(size)
    img, lib = cases.inputs(4096, 4096, size, L, t, lattice=64)
    group = cases.group_of(shape)
    state = G.getGridState(group, 4096, 4096)[0]
    cells = G.cell_records(group, state, 4096, 4096)
    stage(engine, shape, img, lib)
    best, score = engine.generate(cells, state.shape[1], state.shape[0], int(t), *rep)
    grid = np.where(best < L, best, -1)
    sh, masks, nthr = shape_of(shape), shape.masks4(), host_threads()

    prefix = cells[:128]
    assert prefix["grid_y"].max() - prefix["grid_y"].min() == 1          # two full rows of 64
    pos = np.stack([prefix["grid_y"], prefix["grid_x"]], axis=1)
    picks, scores = oracle.picks_given_grid(sh, img, lib, masks, int(t), rep[0], rep[1], grid, pos, threads=nthr)
    got = best[pos[:, 0], pos[:, 1]]
    assert np.array_equal(picks, got), np.flatnonzero(picks != got)
    rel = np.abs(scores - score[pos[:, 0], pos[:, 1]]) / scores
    assert rel.max() <= 1e-4, rel.max()
    # the window was exercised: second-row picks that differ from the unpenalised choice
    plain, _ = engine.generate(cells, state.shape[1], state.shape[0], int(t))
    assert (plain[pos[64:, 0], pos[64:, 1]] != got[64:]).any()

    rng = np.random.default_rng(20261003)
    inner = cells[(cells["grid_y"] >= 8) & (cells["grid_x"] >= 8)]
    sample = inner[rng.choice(len(inner), 32, replace=False)]
    idx, sc = engine.score_cells(sample, int(t), 2)
    spos = np.stack([sample["grid_y"], sample["grid_x"]], axis=1)
    empty = np.full_like(grid, -1)
    first, first_score = oracle.picks_given_grid(sh, img, lib, masks, int(t), 0, 0, empty, spos, threads=nthr,
                                                 early_exit=False)
    assert np.array_equal(first, idx[:, 0])
    assert (np.abs(first_score - sc[:, 0]) <= 1e-4 * first_score).all()
    # runner-up: put the winner in the cell's left neighbour with a repeat value no score reaches
    second = np.empty(32, np.int64)
    second_score = np.empty(32)

    def runner_up(k):
        gy, gx = spos[k]
        g = empty.copy()
        g[gy, gx - 1] = first[k]
        p, s = oracle.picks_given_grid(sh, img, lib, masks, int(t), 1, 1 << 30, g, spos[k:k + 1], early_exit=False)
        second[k], second_score[k] = p[0], s[0]

    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(nthr) as pool:                   # ctypes releases the GIL inside the oracle
        list(pool.map(runner_up, range(32)))
    assert np.array_equal(second, idx[:, 1])
    assert (np.abs(second_score - sc[:, 1]) <= 1e-4 * second_score).all()


def test_config_d_whole_grid_same_on_every_path(engine):
    """Config D, all 4096 cells: default == unpruned == fp64-only, indices and scores."""
    t, L, size, rep = Type.CIEDE2000, 5000, 64, (4, 500)
    shape = cases.square_shape(size)
    img, lib = cases.inputs(4096, 4096, size, L, t, lattice=64)
    group = cases.group_of(shape)
    state = G.getGridState(group, 4096, 4096)[0]
    cells = G.cell_records(group, state, 4096, 4096)
    stage(engine, shape, img, lib)
    assert_same_on_every_path(engine, cells, state.shape[1], state.shape[0], t, rep, state >= 0, L)


def test_config_e_shard_same_on_every_path(engine):
    """Config E's first 1/8 shard (3656 cells of 48 x 48, 10 000 images): the candidate lists of
    the default path == unpruned == fp64-only, indices and scores."""
    t, L, shape, img, lib, cells, n_shard = e_shard_inputs()
    stage(engine, shape, img, lib)
    k = engine.candidates_per_cell(0, 0)
    run = lambda: engine.score_cells(cells, int(t), k)
    idx, sc = run()
    assert engine.timing()["seed_ms"] > 0
    for options in ({"prune": 0}, {"filter": 0}):
        i2, s2 = with_options(engine, options, run)
        assert np.array_equal(i2, idx), (options, int((i2 != idx).sum()))
        assert np.array_equal(s2, sc), options


def test_config_e_one_rank_shard_48px_cells_l10000(oracle, engine):
    """Config E as ONE rank of 8 sees it: the first 1/8 of the 29 241 cells (48x48
    cells, 10 000 images, CIEDE2000), scored with the split API.  The target band
    holding those cells is generated at full width (8192)."""
    t, L, shape, img, lib, cells, n_shard = e_shard_inputs()
    stage(engine, shape, img, lib)
    k = engine.candidates_per_cell(0, 0)
    idx, sc = engine.score_cells(cells, int(t), k)
    assert idx.shape == (n_shard, 2) and (idx >= 0).all() and (idx < L).all()
    assert (sc[:, 0] <= sc[:, 1]).all()
    # oracle on the first 2 cells (band image has the same first rows as the full target)
    want, want_score, done = oracle.generate_best_fits(shape_of(shape), img, lib, shape.masks4(), int(t), max_cells=2)
    for r, i, s in zip(cells[:2], idx[:2, 0], sc[:2, 0]):
        assert i == want[r["grid_y"], r["grid_x"]]
        assert abs(s - want_score[r["grid_y"], r["grid_x"]]) <= 1e-4 * s


@pytest.mark.parametrize("t", [Type.CIEDE2000, Type.RGB_EUCLIDEAN], ids=["ciede2000", "rgb"])
def test_config_e_whole_grid_same_on_every_path(oracle, engine, t):
    """BASELINE config 5 as stated, whole, on one GPU: 8192 x 8192 target, 48 x 48 cells, 171 x 171 grid with the
    clipped last row / column (29 241 cells), 10 000 images, CIEDE2000 -- staged as bench.py --config E stages it
    (u8 images converted on the device) -- and the same images as RGB.  Default path == early exit off == fp64 only:
    every best fit, every score.  (One batch of 29 241 cells x 10 000 images is also where a (cell, tile) grid passes
    2^32 threads: 18.3 M workgroups of the fp64 kernel, 4.6 M 1024-thread ones of the RGB filter.)"""
    import bench
    w = bench.build_workload("E", 1, "strong")
    assert len(w["cells"]) == 171 * 171 and w["L"] == 10000
    engine.set_library_u8(w["lib_u8"], int(t))
    engine.set_target_u8(w["img_u8"], int(t))
    engine.set_masks(w["shape"].masks4())
    cells, gw, gh, rep = w["cells"], w["grid_w"], w["grid_h"], w["rep"]
    run = lambda: engine.generate(cells, gw, gh, int(t), *rep)
    best, score = run()
    tm = engine.timing()
    assert tm["seed_ms"] > 0 and tm["rescore_ms"] > 0 and tm["band_alarms"] == 0
    pos = (cells["grid_y"], cells["grid_x"])
    assert (best[pos] < w["L"]).all()
    for options in ({"prune": 0}, {"filter": 0}):
        b2, s2 = with_options(engine, options, run)
        assert np.array_equal(b2, best), (options, int((b2 != best).sum()))
        assert np.array_equal(s2[pos], score[pos]), options
    # 64 cells drawn at random from the whole grid (clipped last row / column included) against the ORACLE, on the arrays
    # the engine holds (the sample bench.py --config E reports parity on)
    img, lib = engine.get_target(w["rows"], w["cols"]), engine.get_library()
    random_cells_against_the_oracle(oracle, engine, w["shape"], img, lib, t, cells, best, score, w["L"],
                                    rel_tol=1e-4 if t == Type.CIEDE2000 else 1e-12)


def test_more_cells_than_one_batch_holds(oracle, engine):
    """53 248 cells (8 x 8 px on a 2048 x 1664 target) are more than the 32 768 a batch takes: two batches in the
    two-tier path, one resolve over the whole grid with a repeat penalty that reaches across the batch boundary --
    every cell checked by the oracle as the reference's raster loop reaches it; and equal to the fp64-only path."""
    t, L, size = Type.CIEDE2000, 64, 8
    shape = cases.square_shape(size)
    rows, cols = 1664, 2048
    img, lib = cases.inputs(rows, cols, size, L, t, lattice=32)
    group = cases.group_of(shape)
    state = G.getGridState(group, rows, cols)[0]
    cells = G.cell_records(group, state, rows, cols)
    assert len(cells) == 208 * 256 > 32768
    stage(engine, shape, img, lib)
    rep = (2, 40)
    run = lambda: engine.generate(cells, state.shape[1], state.shape[0], int(t), *rep)
    best, score = run()
    assert engine.timing()["score_launches"] >= 2
    b0, s0 = with_options(engine, {"filter": 0}, run)
    assert np.array_equal(b0, best) and np.array_equal(s0[state >= 0], score[state >= 0])
    grid = np.where(best < L, best, -1)
    pos = np.stack([cells["grid_y"], cells["grid_x"]], axis=1)
    picks, _ = oracle.picks_given_grid(shape_of(shape), img, lib, shape.masks4(), int(t), rep[0], rep[1], grid, pos,
                                       threads=host_threads())
    got = best[pos[:, 0], pos[:, 1]]
    assert np.array_equal(np.where(got == L, -1, got), picks)
