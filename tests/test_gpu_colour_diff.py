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


def test_the_parallel_sums_algebraic_form_agrees_with_the_reference_form(engine, oracle):
    """colour_diff.hpp ciede2000_algebraic -- what every PARALLEL-order fp64 sum adds per pair since round 5 (no atan2 / cos per hue,
    about half the instructions) -- against the reference form on the device and against the oracle, over the families where a
    restated form could go wrong: uniform in the box, near-identical pairs, identical pairs (exactly 0, as the reference), exact
    greys on one side and on both, tiny chromas, both sides of the opposite-hue seam and of the h_bar = 0 / 360 wrap (pairs within
    1e-9 of either discontinuity are handed to the reference form by the algebraic one itself), the quarter turn where its arms
    change, saturated blue, the colours of u8 images.  What the engine needs is a TOTAL good to well below the 1e-9 inside which
    candidates are re-summed in the reference's order with the reference's arithmetic: 2e-12 absolute per pair is asserted."""
    rng = np.random.default_rng(2025)
    n = 1 << 19
    lo, hi = np.array([0, -128, -128]), np.array([100, 127, 127])
    two_pi = 2 * np.pi

    def polar(L, c, h):
        return np.stack([L, c * np.cos(h), c * np.sin(h)], axis=1)
    L1, L2 = rng.uniform(0, 100, n), rng.uniform(0, 100, n)
    h1, c1, c2 = rng.uniform(0, two_pi, n), rng.uniform(0, 150, n), rng.uniform(0.5, 150, n)
    off = rng.choice([-1.0, 1.0], n) * 10.0 ** rng.uniform(-12, -0.5, n)
    uni = rng.uniform(lo, hi, (n, 3))
    grey = uni.copy(); grey[:, 1:] = 0
    from mosaicmagnifique_amd import synth
    u8a, u8b = rng.integers(0, 256, (n, 1, 3), dtype=np.uint8), rng.integers(0, 256, (n, 1, 3), dtype=np.uint8)
    families = {
        "uniform": (uni, rng.uniform(lo, hi, (n, 3))),
        "near-identical": (uni, uni + rng.normal(0, 1, (n, 3)) * 10.0 ** rng.uniform(-6, 0, (n, 1))),
        "identical": (uni, uni),
        "grey first": (grey, rng.uniform(lo, hi, (n, 3))),
        "grey both": (grey, grey[::-1]),
        "tiny chroma": (polar(L1, 10.0 ** rng.uniform(-9, -1, n), h1), polar(L2, c2, rng.uniform(0, two_pi, n))),
        "around the seam": (polar(L1, c1, h1), polar(L2, c2, h1 + np.pi + off)),
        "around the wrap of the mean hue": (polar(L1, c2, off), polar(L2, c2, -off * rng.uniform(0.5, 1.5, n))),
        "around the quarter turn": (polar(L1, c1, h1), polar(L2, c2, h1 + np.pi / 2 + off)),
        "saturated blue": (polar(L1, rng.uniform(40, 150, n), np.radians(275) + rng.normal(0, 0.2, n)),
                           polar(L2, rng.uniform(40, 150, n), np.radians(275) + rng.normal(0, 0.3, n))),
        "u8 colours": (synth.bgr_u8_to_lab_f32(u8a).reshape(-1, 3), synth.bgr_u8_to_lab_f32(u8b).reshape(-1, 3)),
    }
    worst = {}
    for name, (a, b) in families.items():
        a, b = np.ascontiguousarray(a, dtype=np.float32), np.ascontiguousarray(b, dtype=np.float32)
        ref = engine.colour_diff_batch(2, a, b)
        engine.set_option("probe_prepared", 3)
        try:
            alg = engine.colour_diff_batch(2, a, b)
        finally:
            engine.set_option("probe_prepared", 0)
        assert np.isfinite(alg).all(), name
        err = np.abs(alg - ref)
        worst[name] = float(err.max())
        assert err.max() <= 2e-12, (name, float(err.max()), a[err.argmax()], b[err.argmax()], float(ref[err.argmax()]))
        if name == "identical":
            assert (alg == 0).all() and (ref == 0).all()
        cpu = oracle.colour_diff_batch(2, a[:4096], b[:4096])
        assert (np.abs(alg[:4096] - cpu) <= 1e-9 * np.maximum(cpu, 1e-3)).all(), name
    print("algebraic fp64 form, worst |difference| to the reference form per family:", worst)


def test_non_finite_pixels_keep_their_nan_through_the_parallel_sums_form(engine, oracle):
    """the algebraic form ended in sqrt(fmax(s, 0)) -- fmax is maxNum, so a NaN radicand became a term of 0 where
    ColourDifference.cpp:42-158 (and the oracle) yield NaN and the candidate can never win.  A pair whose radicand is not finite now
    goes to the reference form: NaN / inf in any channel of either pixel gives what the reference form gives, NaN pattern included."""
    rng = np.random.default_rng(7)
    n = 4096
    a = rng.uniform([0, -100, -100], [100, 100, 100], (n, 3)).astype(np.float32)
    b = rng.uniform([0, -100, -100], [100, 100, 100], (n, 3)).astype(np.float32)
    bad = np.array([np.nan, np.inf, -np.inf], dtype=np.float32)
    for i in range(n):                                   # every (side, channel, kind) combination many times over, plus both sides at once
        side, ch, kind = i % 2, (i // 2) % 3, (i // 6) % 3
        (a if side == 0 else b)[i, ch] = bad[kind]
        if i % 17 == 0:
            (b if side == 0 else a)[i, (ch + 1) % 3] = bad[(kind + 1) % 3]
    ref = engine.colour_diff_batch(2, a, b)
    engine.set_option("probe_prepared", 3)
    try:
        alg = engine.colour_diff_batch(2, a, b)
    finally:
        engine.set_option("probe_prepared", 0)
    assert np.array_equal(np.isnan(alg), np.isnan(ref))
    assert np.array_equal(alg, ref, equal_nan=True)
    assert np.isnan(alg).sum() > n // 2                  # (an inf lightness against a finite one can give inf: whatever the reference form says)
    cpu = oracle.colour_diff_batch(2, a, b)
    assert np.array_equal(np.isnan(alg), np.isnan(cpu))
