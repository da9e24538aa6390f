"""mm_group_*: one process, several engines, the cell-sharded generate with ONE all-gather of the candidate lists
behind the C ABI (the reference is single-device, CUDA/CUDAPhotomosaicGenerator.h:31).  On the one-GPU
box a group over [0] builds a real RCCL communicator (ncclCommInitAll, one rank) and its lists travel through
ncclAllGather; a group that lists device 0 several times gathers through host memory (RCCL refuses a device
twice) -- 2, 3 and 8 shards, the layout an 8-GPU node runs.  Every grid must equal mm_generate's on one engine bit
for bit, scores included, also when short speculative lists force deepening rounds by the shards' owners."""
import numpy as np
import pytest

import cases
from mosaicmagnifique_amd import GridUtility as G
from mosaicmagnifique_amd import synth
from mosaicmagnifique_amd.colourdifference import Type
from mosaicmagnifique_amd.engine import Engine, EngineGroup
from mosaicmagnifique_amd.grid import CELL_DTYPE

pytestmark = pytest.mark.gpu


def scenario(kind):
    if kind == "D":          # the reduced shape of config D: square 64, CIEDE2000, repeat range 4 (+500), smooth target: collisions
        t, rep, size, L, rows, cols = Type.CIEDE2000, (4, 500), 64, 300, 11 * 64, 9 * 64
        shape = cases.square_shape(size)
        img = synth.preprocess(synth.target_u8(rows, cols, lattice=256), t)
    elif kind == "E":        # ... of config E: square 48 with a clipped last row / column, no repeats
        t, rep, size, L, rows, cols = Type.CIEDE2000, (0, 0), 48, 400, 13 * 48 + 20, 12 * 48 + 32
        shape = cases.square_shape(size)
        img = synth.preprocess(synth.target_u8(rows, cols, lattice=64), t)
    else:                    # hexagon cells (offset rows, clipped slivers), RGB, a short repeat window
        t, rep, L = Type.RGB_EUCLIDEAN, (2, 90), 150
        shape = cases.hexagon_shape()
        size, rows, cols = 64, 500, 620
        img = synth.preprocess(synth.target_u8(rows, cols, lattice=128), t)
    lib = synth.preprocess(synth.library_u8(L, size), t)
    group = cases.group_of(shape)
    state = G.getGridState(group, rows, cols)[0]
    cells = G.cell_records(group, state, rows, cols)
    return shape, t, rep, img, lib, state, cells


@pytest.fixture(scope="module")
def single_engine():
    e = Engine(0)
    yield e
    e.close()


@pytest.mark.parametrize("kind", ["D", "E", "H"])
@pytest.mark.parametrize("devices,spec_k", [([0], 4), ([0, 0], 4), ([0, 0, 0], 2), ([0] * 8, 4), ([0] * 8, 2)])
def test_group_equals_one_engine(single_engine, kind, devices, spec_k):
    shape, t, rep, img, lib, state, cells = scenario(kind)
    e = single_engine
    e.set_library(lib); e.set_masks(shape.masks4()); e.set_target(img)
    want, want_score = e.generate(cells, state.shape[1], state.shape[0], int(t), *rep)
    with EngineGroup(devices) as g:
        assert len(g) == len(devices)
        assert g.uses_rccl == (len(devices) == 1)          # a device listed twice: host merge
        g.set_library(lib); g.set_masks(shape.masks4()); g.set_target(img)
        g.set_option("spec_k", spec_k)
        g.set_option("phase_timing", 1)
        got, got_score = g.generate(cells, state.shape[1], state.shape[0], int(t), *rep)
        again, _ = g.generate(cells, state.shape[1], state.shape[0], int(t), *rep)       # buffers are kept between calls
    valid = state >= 0
    assert np.array_equal(got, want) and np.array_equal(again, want)
    assert np.array_equal(got_score[valid], want_score[valid])


def test_group_with_more_devices_than_cells_and_an_empty_grid(single_engine):
    """Shards may be empty (fewer cells than devices; no cell at all)."""
    shape, t, rep, img, lib, state, cells = scenario("E")
    few = cells[:3]
    e = single_engine
    e.set_library(lib); e.set_masks(shape.masks4()); e.set_target(img)
    want, _ = e.generate(few, state.shape[1], state.shape[0], int(t), *rep)
    with EngineGroup([0] * 5) as g:
        g.set_library(lib); g.set_masks(shape.masks4()); g.set_target(img)
        got, _ = g.generate(few, state.shape[1], state.shape[0], int(t), *rep)
        none, _ = g.generate(few[:0], state.shape[1], state.shape[0], int(t), *rep)
    assert np.array_equal(got, want)
    assert (none == len(lib)).all()


def test_group_errors_are_reported_not_thrown():
    from mosaicmagnifique_amd.engine import EngineError
    with pytest.raises(EngineError):
        EngineGroup([0, 99])                       # no such device
    with EngineGroup([0, 0]) as g:
        with pytest.raises(EngineError):           # nothing staged
            g.generate(np.zeros(0, dtype=CELL_DTYPE), 4, 4, 0)
