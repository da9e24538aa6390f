"""BASELINE.json's full-size configurations on the GPU, checked through properties
that do not need a full CPU pass: oracle parity on a raster-order prefix of cells,
library-reversal equivariance, shard/whole agreement."""
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


def test_config_e_one_rank_shard_48px_cells_l10000(oracle, engine):
    """Config E as ONE rank of 8 sees it: the first 1/8 of the 29 241 cells (48x48
    cells, 10 000 images, CIEDE2000), scored with the split API.  The target band
    holding those cells is generated at full width (8192)."""
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
