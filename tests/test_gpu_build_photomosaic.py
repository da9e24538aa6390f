"""buildPhotomosaic on the device, byte-exact against the oracle's
restatement of PhotomosaicGeneratorBase.cpp:110-207 (raster-order painting through the
flipped masks: the later cell wins where shapes overlap)."""
import numpy as np
import pytest

import cases
from mosaicmagnifique_amd import GridUtility as G
from mosaicmagnifique_amd import HIPPhotomosaicGenerator, synth
from mosaicmagnifique_amd.colourdifference import Type
from mosaicmagnifique_amd.engine import Engine, EngineError
from oracle_ffi import shape_of

pytestmark = pytest.mark.gpu

SHAPES = {"square16": lambda: cases.square_shape(16), "hexagon64": cases.hexagon_shape,
          "flippy24": cases.flippy_shape}


@pytest.mark.parametrize("shape_name", SHAPES)
def test_generate_then_build_matches_oracle(oracle, shape_name):
    shape = SHAPES[shape_name]()
    size = shape.getSize()
    rows, cols, n_lib = 5 * size + 7, 6 * size + 3, 50
    img_u8 = synth.target_u8(rows, cols, lattice=2 * size)
    lib_u8 = synth.library_u8(n_lib, size)
    group = cases.group_of(shape)
    gen = HIPPhotomosaicGenerator(0)
    gen.setMainImage(img_u8); gen.setLibrary(lib_u8); gen.setColourDifference(Type.CIE76)
    gen.setCellGroup(group); gen.setGridState(G.getGridState(group, rows, cols)); gen.setRepeat(1, 30)
    assert gen.generateBestFits(), gen.lastError
    fits = gen.getBestFits()[0]
    got = gen.buildPhotomosaic((10, 20, 30, 40))
    want = oracle.build_photomosaic(shape_of(shape), rows, cols, lib_u8, shape.masks4(), fits, (10, 20, 30, 40))
    assert got.dtype == np.uint8 and got.shape == (rows, cols, 4)
    assert np.array_equal(got, want)
    # square cells tile the image completely; offset / masked shapes leave background
    covered = (got[..., 3] == 255).mean()
    assert covered == 1.0 if shape_name == "square16" else 0.3 < covered <= 1.0


def test_cells_without_a_value_are_skipped_and_overlaps_take_the_later_cell(oracle, engine):
    shape = cases.hexagon_shape()                    # rows overlap: 64-px cells every 48 px
    group = cases.group_of(shape)
    rows, cols, n_lib = 300, 330, 7
    lib_u8 = synth.library_u8(n_lib, 64)
    state = G.getGridState(group, rows, cols)[0]
    cells = G.cell_records(group, state, rows, cols)
    fits = np.where(state >= 0, (np.arange(state.size).reshape(state.shape) * 5) % n_lib, -1)
    fits[3, 3] = -1                                   # std::nullopt
    fits[4, 4] = n_lib                                # the device path's sentinel
    engine.set_library_bgr(lib_u8)
    engine.set_masks(shape.masks4())
    got = engine.build_photomosaic(cells, state.shape[1], state.shape[0], fits, rows, cols, (1, 2, 3, 4))
    want = oracle.build_photomosaic(shape_of(shape), rows, cols, lib_u8, shape.masks4(),
                                    np.where(fits >= n_lib, -1, fits), (1, 2, 3, 4))
    assert np.array_equal(got, want)
    assert engine.timing()["build_ms"] > 0


def test_full_size_config_d_mosaic(oracle, engine):
    """4096 x 4096, 64 x 64 square cells, 5000-image library: every pixel comes from the
    chosen tile (spot-checked against the oracle on the whole image)."""
    shape = cases.square_shape(64)
    group = cases.group_of(shape)
    rows = cols = 4096
    lib_u8 = synth.library_u8(5000, 64)
    state = G.getGridState(group, rows, cols)[0]
    cells = G.cell_records(group, state, rows, cols)
    rng = np.random.default_rng(2)
    fits = np.where(state >= 0, rng.integers(0, 5000, state.shape), -1)
    engine.set_library_bgr(lib_u8)
    engine.set_masks(shape.masks4())
    got = engine.build_photomosaic(cells, state.shape[1], state.shape[0], fits, rows, cols)
    want = oracle.build_photomosaic(shape_of(shape), rows, cols, lib_u8, shape.masks4(), fits)
    assert np.array_equal(got, want)


def test_build_errors():
    e = Engine(0)
    rec = np.zeros(1, dtype=G.CELL_DTYPE); rec["row_max"] = rec["col_max"] = 8; rec["grid_x"] = rec["grid_y"] = 1
    with pytest.raises(EngineError) as err:
        e.build_photomosaic(rec, 3, 3, np.zeros((3, 3), np.int64), 8, 8)
    assert err.value.status == 5                      # nothing staged
    e.set_library_bgr(np.zeros((2, 8, 8, 3), np.uint8)); e.set_masks(np.full((4, 8, 8), 255, np.uint8))
    bad = rec.copy(); bad["x"] = 3
    with pytest.raises(EngineError) as err:
        e.build_photomosaic(bad, 3, 3, np.zeros((3, 3), np.int64), 8, 8)
    assert err.value.status == 2
    out = e.build_photomosaic(rec, 3, 3, np.ones((3, 3), np.int64), 8, 8, (9, 9, 9, 9))
    assert (out[..., :3] == 0).all() and (out[..., 3] == 255).all()
    e.close()
