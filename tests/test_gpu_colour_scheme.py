"""Colour-scheme variants (ColourScheme.cpp:44-177) and what the generators do with them.

OpenCV's cvtColor is absent from the reference checkout and no reference test holds
vectors for BGR2HSV_FULL / HSV2BGR_FULL: the arithmetic is specified (csrc/kernels.hpp
hue_rotate_px) -- parity unpinned at this step.  What IS checked: device == oracle == the
numpy restatement byte for byte, the properties the construction must have, and that a
generator with a scheme set returns the reference's result, i.e. the best fits against
the scheme's LAST variant (the reference's variant cells share one buffer,
PhotomosaicGeneratorBase.cpp:310-312)."""
import numpy as np
import pytest

import cases
from mosaicmagnifique_amd import ColourScheme, GridUtility as G, HIPPhotomosaicGenerator, synth
from mosaicmagnifique_amd.colourdifference import Type
from oracle_ffi import shape_of


def sample_image():
    rng = np.random.default_rng(5)
    img = rng.integers(0, 256, (257, 263, 3), dtype=np.uint8)
    corners = np.array([[0, 0, 0], [255, 255, 255], [255, 0, 0], [0, 255, 0], [0, 0, 255], [128, 128, 128],
                        [255, 255, 0], [0, 255, 255], [255, 0, 255], [1, 0, 0], [254, 255, 255], [10, 10, 11],
                        [200, 100, 100], [100, 200, 100], [100, 100, 200], [7, 7, 7]], dtype=np.uint8)
    img[0, :16] = corners
    g = np.arange(256, dtype=np.uint8)
    img[1, :256] = np.stack([g, g, g], axis=1)                  # greys: S == 0 branch
    img[2, :256] = np.stack([g, g[::-1], np.full(256, 255, np.uint8)], axis=1)
    return img


def test_numpy_restatement_equals_oracle(oracle):
    img = sample_image()
    for scheme, rots in ColourScheme.ROTATIONS.items():
        for d in rots:
            assert np.array_equal(ColourScheme.hue_rotate_u8(img, d), oracle.hue_rotate_u8(img, d)), (scheme, d)


def test_rotation_properties(oracle):
    img = sample_image()
    assert np.array_equal(oracle.hue_rotate_u8(img, 0.0), img)                # no rotation: the round trip is the identity on u8
    grey = img[1:2, :256]
    for d in (30.0, 180.0, 270.0):
        assert np.array_equal(oracle.hue_rotate_u8(grey, d), grey)            # greys have no hue
        out = oracle.hue_rotate_u8(img, d).astype(np.int32)
        assert np.abs(out.max(-1) - img.astype(np.int32).max(-1)).max() <= 1  # V = max(B,G,R) survives
        assert np.abs(out.min(-1) - img.astype(np.int32).min(-1)).max() <= 1  # and so does V (1 - S)
    pure = np.array([[[0, 0, 255]]], dtype=np.uint8)                          # red, BGR
    assert oracle.hue_rotate_u8(pure, 120.0).tolist() == [[[0, 255, 0]]]      # -> green
    assert oracle.hue_rotate_u8(pure, 240.0).tolist() == [[[255, 0, 0]]]      # -> blue
    assert oracle.hue_rotate_u8(pure, 180.0).tolist() == [[[255, 255, 0]]]    # -> cyan


def test_scheme_tables():
    assert [len(ColourScheme.getFunction(t)(np.zeros((2, 2, 3), np.uint8))) for t in range(6)] == [1, 2, 3, 3, 4, 4]
    assert ColourScheme.strToEnum("Triadic") == ColourScheme.Type.TRIADIC
    assert ColourScheme.strToEnum("nope") == ColourScheme.Type.MAX
    with pytest.raises(ValueError):
        ColourScheme.getFunction(ColourScheme.Type.MAX)


@pytest.mark.gpu
def test_device_rotation_is_byte_exact(oracle, engine):
    img = sample_image()
    for d in (30.0, 60.0, 90.0, 120.0, 150.0, 180.0, 210.0, 240.0, 270.0, 359.5, 0.0):
        assert np.array_equal(engine.hue_rotate_u8(img, d), oracle.hue_rotate_u8(img, d)), d
    # every u8 colour through one rotation
    c = np.arange(1 << 24, dtype=np.uint32)
    cube = np.stack([c & 255, (c >> 8) & 255, c >> 16], axis=1).astype(np.uint8).reshape(4096, 4096, 3)
    assert np.array_equal(engine.hue_rotate_u8(cube, 150.0), oracle.hue_rotate_u8(cube, 150.0))


@pytest.mark.gpu
@pytest.mark.parametrize("scheme", [ColourScheme.Type.COMPLEMENTARY, ColourScheme.Type.TRIADIC,
                                    ColourScheme.Type.COMPOUND, ColourScheme.Type.TETRADIC, ColourScheme.Type.ANALAGOUS])
@pytest.mark.parametrize("t", [Type.RGB_EUCLIDEAN, Type.CIEDE2000])
def test_generator_with_scheme_equals_reference_behaviour(oracle, scheme, t):
    shape = cases.hexagon_shape() if t == Type.CIEDE2000 else cases.square_shape(16)
    size = shape.getSize()
    rows, cols, n_lib = 3 * size + 11, 4 * size + 5, 48
    img = synth.target_u8(rows, cols, lattice=size)
    lib = synth.library_u8(n_lib, size)
    group = cases.group_of(shape)
    gen = HIPPhotomosaicGenerator(0)
    gen.setMainImage(img); gen.setLibrary(lib); gen.setColourDifference(t); gen.setColourScheme(scheme)
    gen.setCellGroup(group); gen.setGridState(G.getGridState(group, rows, cols)); gen.setRepeat(2, 200)
    assert gen.generateBestFits(), gen.lastError
    variants = gen.mainImageVariants()
    assert len(variants) == 1 + len(ColourScheme.ROTATIONS[scheme]) and np.array_equal(variants[0], img)
    for v, d in zip(variants[1:], ColourScheme.ROTATIONS[scheme]):
        assert np.array_equal(v, oracle.hue_rotate_u8(img, d))
    # the reference's scan sees the last variant in every variant's cell
    want, want_score, _ = oracle.generate_best_fits(shape_of(shape), synth.preprocess(variants[-1], t),
                                                    synth.preprocess(lib, t), shape.masks4(), int(t), 2, 200)
    assert np.array_equal(gen.getBestFits()[0], want)
    valid = want >= 0
    got_score = gen.getBestScores()[0]
    assert np.allclose(got_score[valid], want_score[valid], rtol=1e-4 if t == Type.CIEDE2000 else 1e-12, atol=0)
    with pytest.raises(ValueError):
        gen.setColourScheme(ColourScheme.Type.MAX)
