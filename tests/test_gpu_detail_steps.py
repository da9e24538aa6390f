"""Detail levels, size steps and pre-processing on the device: the
cases the reference's own generator tests run most -- CreateCellGroup(128, 0, 50) and
(128, 1, 50), test/tst_CUDAGenerator.h:92-181, tst_Generator.h:75-86 -- at small sizes,
against the oracle fed with the SAME pre-processed arrays the engine holds."""
import numpy as np
import pytest

import cases
from mosaicmagnifique_amd import GridUtility as G
from mosaicmagnifique_amd import HIPPhotomosaicGenerator, imageutil, synth
from mosaicmagnifique_amd.cellshape import CellGroup
from mosaicmagnifique_amd.colourdifference import Type
from oracle_ffi import shape_of

pytestmark = pytest.mark.gpu

TYPES = [Type.RGB_EUCLIDEAN, Type.CIE76, Type.CIEDE2000]


def test_lab_conversion_of_every_u8_colour(engine):
    """preprocess on the device == synth.bgr_u8_to_lab_f32 for ALL 2^24 colours (within
    1 ulp of f32: the device and glibc cbrt may round differently) and exactly for RGB."""
    worst = 0.0
    differ = 0
    for b in range(0, 256, 64):
        blue = np.arange(b, b + 64, dtype=np.uint8)
        img = np.stack(np.meshgrid(blue, np.arange(256, dtype=np.uint8), np.arange(256, dtype=np.uint8),
                                   indexing="ij"), axis=-1).reshape(64 * 256, 256, 3)
        engine.set_target_u8(img, int(Type.CIEDE2000))
        got = engine.get_target(*img.shape[:2])
        want = synth.bgr_u8_to_lab_f32(img)
        err = np.abs(got.astype(np.float64) - want.astype(np.float64))
        worst = max(worst, float(err.max()))
        differ += int((got != want).sum())
        assert (err <= 1.2e-7 * np.maximum(np.abs(want), 1.0) * 8).all()
        engine.set_target_u8(img, int(Type.RGB_EUCLIDEAN))
        assert np.array_equal(engine.get_target(*img.shape[:2]), img.astype(np.float32))
    print("Lab: max abs difference %.3g, %d of %d values differ in the last place" % (worst, differ, 3 << 24))
    assert differ < 1000


@pytest.mark.parametrize("f", [1, 2, 4, 3])
def test_library_u8_resize_then_convert(oracle, engine, f):
    """preprocessLibraryImages: INTER_AREA on the u8 image, then the colour conversion."""
    lib = synth.library_u8(40, 24)
    for t in (Type.RGB_EUCLIDEAN, Type.CIE76):
        engine.set_library_u8(lib, int(t), shrink=f)
        got = engine.get_library()
        small = np.stack([imageutil.shrink_u8(im, f) for im in lib])
        assert np.array_equal(small, np.stack([oracle.resize_area_u8(im, 24 // f, 24 // f) for im in lib]))
        want = synth.preprocess(small, t)
        assert got.shape == want.shape
        assert np.abs(got - want).max() <= (0 if t == Type.RGB_EUCLIDEAN else 1e-5)


@pytest.mark.parametrize("f", [2, 3])
def test_shrink_library_is_the_specified_box_mean(oracle, engine, f):
    lib = synth.preprocess(synth.library_u8(30, 24), Type.CIE76)
    engine.set_library(lib)
    engine.shrink_library(f)
    got = engine.get_library()
    assert np.array_equal(got, oracle.shrink_library_f32(lib, f))
    assert np.array_equal(got, np.stack([imageutil.shrink_f32(im, f) for im in lib]))


def run_steps(oracle, shape, img_u8, lib_u8, t, detail, steps, rep=(0, 0), spec_k=None):
    group = CellGroup(shape)
    group.setDetail(detail)
    group.setSizeSteps(steps)
    state = G.getGridState(group, img_u8.shape[0], img_u8.shape[1], mainImage=img_u8)
    gen = HIPPhotomosaicGenerator(0)
    if spec_k is not None:
        gen.m_engine.set_option("spec_k", spec_k)
    gen.setMainImage(img_u8)
    gen.setLibrary(lib_u8)
    gen.setColourDifference(t)
    gen.setCellGroup(group)
    gen.setGridState(state)
    gen.setRepeat(*rep)
    main = gen.preprocessMainImage()
    lib = gen.preprocessLibraryImages()
    assert lib.shape[1] == group.getCellSize(0, True)
    assert gen.generateBestFits(), gen.lastError
    fits, scores = gen.getBestFits(), gen.getBestScores()
    assert len(fits) == len(state)
    for step in range(len(state)):
        if step > 0:
            lib = oracle.shrink_library_f32(lib, 2)          # batchResizeMat between the steps
        d = group.getCell(step, True)
        want, want_score, _ = oracle.generate_step(
            shape_of(group.getCell(step)), d.getSize(), group.getDetail(), main, lib, d.masks4(),
            int(t), rep[0], rep[1], state=state[step])
        assert np.array_equal(fits[step], want), "step %d" % step
        valid = want >= 0
        tol = 1e-4 if t == Type.CIEDE2000 else 1e-12
        assert (np.abs(scores[step][valid] - want_score[valid]) <= tol * np.abs(want_score[valid])).all()
    return gen, state


@pytest.mark.parametrize("rep", [(0, 0), (2, 400)], ids=["norepeat", "repeat"])
@pytest.mark.parametrize("t", TYPES)
def test_detail_50(oracle, t, rep):
    """Detail 50 %: cells box-averaged 2x2 from the image (zero-filled where clipped), library
    resized on the u8 images.  Ragged image so the clipped bounds truncate (odd widths)."""
    img = synth.target_u8(205, 233, lattice=48)
    lib = synth.library_u8(120, 32)
    gen, _ = run_steps(oracle, cases.square_shape(32), img, lib, t, 50, 0, rep)
    assert gen.m_engine.size == 16


def test_detail_25_flipped_offset_shape(oracle):
    img = synth.target_u8(150, 170, lattice=40)
    lib = synth.library_u8(90, 24)
    run_steps(oracle, cases.flippy_shape(24), img, lib, Type.CIEDE2000, 25, 0, (2, 300))


@pytest.mark.parametrize("detail", [100, 50])
@pytest.mark.parametrize("t", TYPES)
def test_size_steps(oracle, t, detail):
    """Size steps 1 and 2 with the entropy subdivision: busy cells (entropy >= 5.6) are
    handed to the half-size step, whose library is the halved one."""
    img = synth.target_u8(200, 264, lattice=64)
    # make the right half busy: strong noise drives the masked grey entropy over 5.6
    noise = (synth.mix(0x77, np.arange(200)[:, None], np.arange(132)[None, :]) % np.uint32(200)).astype(np.int16) - 100
    img[:, 132:] = np.clip(img[:, 132:].astype(np.int16) + noise[..., None], 0, 255).astype(np.uint8)
    lib = synth.library_u8(100, 32)
    gen, state = run_steps(oracle, cases.square_shape(32), img, lib, t, detail, 2, (2, 500), spec_k=3)
    assert len(state) >= 2
    assert (state[0] >= 0).any() and (state[1] >= 0).any()
    # a cell split at step 0 has no value there
    assert (state[0] < 0).sum() > (G.getGridState(CellGroup(cases.square_shape(32)), 200, 264)[0] < 0).sum()


def test_build_photomosaic_over_size_steps(oracle):
    """buildPhotomosaic with size steps: the half-size cells of step 1 fill only what step 0 left."""
    img = synth.target_u8(200, 264, lattice=64)
    noise = (synth.mix(0x77, np.arange(200)[:, None], np.arange(132)[None, :]) % np.uint32(200)).astype(np.int16) - 100
    img[:, 132:] = np.clip(img[:, 132:].astype(np.int16) + noise[..., None], 0, 255).astype(np.uint8)
    lib = synth.library_u8(100, 32)
    for shape in (cases.square_shape(32), cases.hexagon_shape().resized(32)):
        lib_s = synth.library_u8(100, shape.getSize())
        gen, state = run_steps(oracle, shape, img, lib_s, Type.CIE76, 50, 1, (1, 200))
        assert len(state) == 2
        group = gen.getCellGroup()
        fits = gen.getBestFits()
        bg = (10, 20, 30, 0)
        got = gen.buildPhotomosaic(bg)
        want = oracle.build_photomosaic_steps([shape_of(group.getCell(s)) for s in range(2)], 200, 264, lib_s,
                                              [group.getCell(s).masks4() for s in range(2)], fits, bg)
        assert np.array_equal(got, want)
        assert (got[..., 3] == 255).any() and got.shape == (200, 264, 4)
        # generating again after building (masks and library were re-staged for painting) still works
        gen.setGridState(state)
        assert gen.generateBestFits(), gen.lastError
        assert all(np.array_equal(a, b) for a, b in zip(gen.getBestFits(), fits))


def test_what_is_still_refused(engine):
    group = CellGroup(cases.square_shape(30))
    group.setDetail(40)                       # 30 -> 12: ratio 2.5, the general resize
    assert group.cellScale(0) == 0
    group.setDetail(50)
    assert group.cellScale(0) == 2
    from mosaicmagnifique_amd.engine import EngineError
    with pytest.raises(EngineError):           # a detail cell larger than the cell: INTER_CUBIC, not on the path
        engine.set_cell_resize(16, 24)
    engine.set_library(synth.preprocess(synth.library_u8(4, 15), Type.RGB_EUCLIDEAN))
    with pytest.raises(EngineError):           # halving an odd library size between size steps
        engine.shrink_library(2)


@pytest.mark.parametrize("detail_size", [23, 18, 17, 7, 1])
def test_library_u8_general_resize_then_convert(oracle, engine, detail_size):
    """A detail level whose cell-size ratio is not whole (24 -> 23 / 18 / 17 / 7 / 1 px): cv::resize's
    general INTER_AREA on the u8 images (computeResizeAreaTab + ResizeArea_Invoker), then the
    colour conversion.  Device == oracle == numpy restatement, byte for byte before conversion."""
    lib = synth.library_u8(40, 24)
    small = np.stack([oracle.resize_area_u8(im, detail_size, detail_size) for im in lib])
    assert np.array_equal(small, np.stack([imageutil.resize_area_u8(im, detail_size, detail_size) for im in lib]))
    for t in (Type.RGB_EUCLIDEAN, Type.CIE76):
        engine.set_library_u8_sized(lib, int(t), detail_size)
        got = engine.get_library()
        want = synth.preprocess(small, t)
        assert got.shape == want.shape
        assert np.abs(got - want).max() <= (0 if t == Type.RGB_EUCLIDEAN else 1e-5)


@pytest.mark.parametrize("shape_name,detail,t,rep", [
    ("square32", 75, Type.RGB_EUCLIDEAN, (0, 0)),        # 32 -> 24: ratio 4/3
    ("square32", 60, Type.CIEDE2000, (2, 400)),           # 32 -> 19
    ("square32", 30, Type.CIE76, (1, 100)),               # 32 -> 9
    ("hexagon", 75, Type.CIEDE2000, (2, 300)),            # 64 -> 48, overlapping rows, clipped x = -1 / y = -1 cells
    ("flippy", 70, Type.CIE76, (1, 50)),                  # 24 -> 16 (int(16.8)), all four mask variants differ
    ("square32", 97, Type.RGB_EUCLIDEAN, (0, 0)),         # 32 -> 31: almost every detail pixel straddles two
])
def test_fractional_detail(oracle, shape_name, detail, t, rep):
    """Detail levels whose cell-size ratio is not a whole number: getCellAt's resize of the zero-filled
    cell is the general INTER_AREA (f32, table order), the bound is scaled with int truncation, the
    mask comes from CellShape::resized.  Indices bit-exact and scores within tolerance against the
    oracle fed the same pre-processed arrays; ragged image sizes so clipped cells are resized too."""
    shape = {"square32": lambda: cases.square_shape(32), "hexagon": cases.hexagon_shape,
             "flippy": lambda: cases.flippy_shape(24)}[shape_name]()
    size = shape.getSize()
    img = synth.target_u8(4 * size + size // 3, 5 * size + size // 2, lattice=2 * size)
    lib = synth.library_u8(70, size)
    gen, _ = run_steps(oracle, shape, img, lib, t, detail, 0, rep)
    assert gen.m_engine.size == max(int(size * detail / 100.0), 1)


def test_fractional_detail_with_a_size_step(oracle):
    """Detail 75 % with one size step where the sizes stay consistent (32 -> 24, 16 -> 12; the library
    halves 24 -> 12): both steps use the general resize of the cell."""
    img = synth.target_u8(150, 200, lattice=64)
    noise = (synth.mix(0x91, np.arange(150)[:, None], np.arange(100)[None, :]) % np.uint32(200)).astype(np.int16) - 100
    img[:, 100:] = np.clip(img[:, 100:].astype(np.int16) + noise[..., None], 0, 255).astype(np.uint8)
    lib = synth.library_u8(60, 32)
    gen, state = run_steps(oracle, cases.square_shape(32), img, lib, Type.CIEDE2000, 75, 1, (1, 200))
    assert len(state) == 2 and (state[1] >= 0).any()
