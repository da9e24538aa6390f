"""Per-pair colour differences through the C ABI (mm_colour_diff_batch) against the
reference's KATs and against the oracle -- the counterparts of the reference's
*_CUDA, *_CPUvsCUDA and *_CPUvsBatchCUDA suites (test/tst_ColourDifference.h:233-466)."""
import json
import os

import numpy as np
import pytest

from mosaicmagnifique_amd.engine import EngineError

pytestmark = pytest.mark.gpu

KATS = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "colour_diff_kats.json")))


@pytest.mark.parametrize("name", ["RGBEuclidean", "CIE76", "CIEDE2000"])
def test_kats_on_device(engine, oracle, name):
    """tst_ColourDifference.h:233-309, at the reference's tolerances.  Inputs pass
    through f32 exactly as the reference's 1x1 CV_32FC3 images do.  One Sharma pair
    -- (50, 2.49, -0.001) vs (50, -2.49, 0.001), the |h1' - h2'| == 180 degree knife
    edge -- changes branch when its inputs are rounded to f32 (7.2195 instead of
    7.1792) on the CPU too; such pairs are held to the CPU value on the same f32
    inputs instead of the table."""
    case = KATS["cases"][name]
    a = np.array([p["first"] for p in case["pairs"]], dtype=np.float32)
    b = np.array([p["second"] for p in case["pairs"]], dtype=np.float32)
    want = np.array([p["difference"] for p in case["pairs"]])
    got = engine.colour_diff_batch(case["type"], a, b)
    cpu = oracle.colour_diff_batch(case["type"], a, b)
    assert (np.abs(got - cpu) <= 1e-9 * np.maximum(cpu, 1e-3)).all()
    tol = max(case["tolerance"], 1e-4 if name == "CIEDE2000" else 1e-5)   # f32 inputs
    stable = np.abs(cpu - want) <= tol
    assert stable.sum() >= len(want) - 1
    assert np.abs(got - want)[stable].max() <= tol


def test_rgb_is_bit_exact_vs_oracle(engine, oracle):
    """u8-valued inputs: every intermediate is an exact integer and the fp64 sqrt is
    correctly rounded, so the device value must EQUAL the CPU value."""
    rng = np.random.default_rng(11)
    a = rng.integers(0, 256, (1 << 20, 3)).astype(np.float32)
    b = rng.integers(0, 256, (1 << 20, 3)).astype(np.float32)
    assert np.array_equal(engine.colour_diff_batch(0, a, b), oracle.colour_diff_batch(0, a, b))


def test_cie76_is_bit_exact_vs_oracle(engine, oracle):
    """Lab f32 inputs, no FMA contraction on either side: still bit-identical."""
    rng = np.random.default_rng(12)
    lo, hi = np.array([0, -128, -128]), np.array([100, 127, 127])
    a = rng.uniform(lo, hi, (1 << 18, 3)).astype(np.float32)
    b = rng.uniform(lo, hi, (1 << 18, 3)).astype(np.float32)
    assert np.array_equal(engine.colour_diff_batch(1, a, b), oracle.colour_diff_batch(1, a, b))


def test_ciede2000_random_lab_vs_oracle(engine, oracle):
    """*_CPUvsCUDA (tst_ColourDifference.h:315-395): 2^14 random Lab pairs within 1e-4
    absolute; here also 1e-9 relative (both sides are all-fp64, only the libm
    roundings of atan2/sin/cos/exp differ)."""
    rng = np.random.default_rng(13)
    lo, hi = np.array([0, -128, -128]), np.array([100, 127, 127])
    a = rng.uniform(lo, hi, (1 << 14, 3)).astype(np.float32)
    b = rng.uniform(lo, hi, (1 << 14, 3)).astype(np.float32)
    got, want = engine.colour_diff_batch(2, a, b), oracle.colour_diff_batch(2, a, b)
    assert np.abs(got - want).max() <= 1e-4
    assert (np.abs(got - want) <= 1e-9 * np.maximum(want, 1e-3)).all()


def test_ciede2000_batch_image_and_special_values(engine, oracle):
    """*_CPUvsBatchCUDA (tst_ColourDifference.h:401-466): a 1000x1000 image in one call;
    plus greys (zero chroma), identical pixels and hue wrap-around pairs."""
    rng = np.random.default_rng(14)
    lo, hi = np.array([0, -128, -128]), np.array([100, 127, 127])
    a = rng.uniform(lo, hi, (1000 * 1000, 3)).astype(np.float32)
    b = rng.uniform(lo, hi, (1000 * 1000, 3)).astype(np.float32)
    a[:1000, 1:] = 0                      # grey main pixel
    b[500:1500, 1:] = 0                   # grey library pixel (overlap: both grey)
    b[2000:3000] = a[2000:3000]           # identical
    a[3000:4000, 2] = -1e-3               # hue just below 2*pi vs just above 0
    b[3000:4000, 2] = 1e-3
    a[3000:4000, 1] = b[3000:4000, 1] = 40.0
    got, want = engine.colour_diff_batch(2, a, b), oracle.colour_diff_batch(2, a, b)
    assert np.isfinite(got).all()
    assert (got[2000:3000] == 0).all()
    assert np.abs(got - want).max() <= 1e-4
    assert (np.abs(got - want) <= 1e-9 * np.maximum(want, 1e-3)).all()


def test_empty_batch_and_bad_type(engine):
    assert engine.colour_diff_batch(0, np.zeros((0, 3)), np.zeros((0, 3))).shape == (0,)
    with pytest.raises(EngineError) as err:
        engine.colour_diff_batch(3, np.zeros((1, 3)), np.zeros((1, 3)))     # Type::MAX
    assert err.value.status == 2 and "No function for given type" in str(err.value)
