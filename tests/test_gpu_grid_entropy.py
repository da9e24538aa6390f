"""The entropy subdivision of the grid on the device (mm_cell_entropy; GridGenerator::findCellState,
Grid/GridGenerator.cpp:113-193, Other/ImageUtility.cpp:189-242): grid states over size steps bit-equal to the
oracle's C restatement (orc_grid_state_steps) on the matrix tests/test_grid_steps.py holds the host mirrors to,
plus detail levels whose resize is the general INTER_AREA (75 %, 30 %) and clipped border cells, whose visible part
shrinks by a different ratio per axis; per-cell entropies equal the host's to the last bits of log2."""
import numpy as np
import pytest

import cases
from mosaicmagnifique_amd import GridUtility as G
from mosaicmagnifique_amd import HIPPhotomosaicGenerator, imageutil, synth
from mosaicmagnifique_amd.cellshape import CellGroup
from oracle_ffi import shape_of
from test_grid_steps import busy_image

pytestmark = pytest.mark.gpu

SHAPES = {"square32": lambda: cases.square_shape(32), "flippy24": cases.flippy_shape, "hexagon64": cases.hexagon_shape}


@pytest.mark.parametrize("detail", [100, 75, 50, 30])
@pytest.mark.parametrize("shape_name", sorted(SHAPES))
def test_grid_state_over_size_steps_on_the_device(oracle, engine, shape_name, detail):
    shape = SHAPES[shape_name]()
    group = CellGroup(shape)
    group.setDetail(detail)
    group.setSizeSteps(2)
    img = busy_image(150, 212)
    got = G.getGridState(group, 0, 0, mainImage=img, engine=engine)
    n_steps = 3
    want = oracle.grid_state_steps([shape_of(group.getCell(s)) for s in range(n_steps)],
                                   [group.getCellSize(s, True) for s in range(n_steps)],
                                   [group.getCell(s, True).masks4() for s in range(n_steps)],
                                   group.getDetail(), img, 150, 212)
    assert len(got) == len(want) and (len(got) >= 2 or detail <= 30)     # (tiny detail masks cannot reach 5.6 bits)
    for a, b in zip(got, want):
        assert np.array_equal(a, b)
    host = G.getGridState(group, 0, 0, mainImage=img)              # the Python mirror, cell by cell on the CPU
    assert all(np.array_equal(a, b) for a, b in zip(got, host))


def test_entropies_equal_the_hosts(engine):
    """Every cell of a step, clipped border cells included (their visible part is resized by a different ratio per
    axis): the device's entropy against imageutil.calculateEntropy on the host-resized cell."""
    shape = cases.hexagon_shape()
    group = CellGroup(shape)
    group.setDetail(60)                      # 64 -> 38 px: the general INTER_AREA
    group.setSizeSteps(1)
    img = busy_image(230, 300, seed=11)
    rows, cols = img.shape[:2]
    state = G.getGridState(group, rows, cols)[0]
    recs = G.cell_records(group, state, rows, cols, 0)
    detail = group.getCell(0, True)
    engine.set_main_image_bgr(img)
    engine.set_masks(detail.masks4())
    exceeded, ent = engine.cell_entropy(recs, shape.getSize(), want_entropy=True)
    masks = detail.masks4()
    size = shape.getSize()
    worst = 0.0
    for c, e_dev, x_dev in zip(recs, ent, exceeded):
        y0, y1 = min(max(int(c["y"]), 0), rows), min(max(int(c["y"]) + size, 0), rows)
        x0, x1 = min(max(int(c["x"]), 0), cols), min(max(int(c["x"]) + size, 0), cols)
        mask = masks[c["mask_id"]][c["row_min"]:c["row_max"], c["col_min"]:c["col_max"]]
        cell = img[y0:y1, x0:x1]
        if cell.shape[:2] != mask.shape:
            cell = imageutil.resize_area_u8(cell, mask.shape[0], mask.shape[1])
        want = imageutil.calculateEntropy(cell, mask)
        worst = max(worst, abs(want - e_dev))
        assert abs(want - e_dev) <= 1e-12
        assert bool(x_dev) == (want >= imageutil.MAX_ENTROPY * 0.7) or abs(want - imageutil.MAX_ENTROPY * 0.7) < 1e-12
    assert (exceeded != 0).any() and (exceeded == 0).any()
    print("max |device - host| entropy: %.3g over %d cells" % (worst, len(recs)))


def test_generator_over_device_grid_state(oracle):
    """The generator's own helper: grid state on its device, then generateBestFits over the steps it names."""
    img = busy_image(200, 264, seed=3)
    lib = synth.library_u8(80, 32)
    group = CellGroup(cases.square_shape(32))
    group.setDetail(50)
    group.setSizeSteps(1)
    gen = HIPPhotomosaicGenerator(0)
    gen.setMainImage(img); gen.setLibrary(lib); gen.setCellGroup(group)
    state = gen.gridStateOnDevice()
    host = G.getGridState(group, 0, 0, mainImage=img)
    assert len(state) == len(host) == 2 and all(np.array_equal(a, b) for a, b in zip(state, host))
    gen.setGridState(state)
    assert gen.generateBestFits(), gen.lastError
    assert all((f[s >= 0] >= 0).all() for f, s in zip(gen.getBestFits(), state))


def test_what_the_entropy_entry_refuses(engine):
    from mosaicmagnifique_amd.engine import EngineError
    shape = cases.square_shape(32)
    group = CellGroup(shape)
    img = busy_image(64, 64)
    engine.set_main_image_bgr(img)
    engine.set_masks(shape.masks4())
    recs = G.cell_records(group, G.getGridState(group, 64, 64)[0], 64, 64)
    with pytest.raises(EngineError):
        engine.cell_entropy(recs, 16)                  # detail masks larger than the normal cell: an enlargement
    assert len(engine.cell_entropy(recs[:0], 32)) == 0
