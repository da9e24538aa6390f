"""The bands the engine trusts its two approximate CIEDE2000 forms with, held against the worst-case bound of
tools/filter_error_bound.py -- the form's own sequence of operations evaluated in float64 with a bound on what the
device's roundings can do carried beside every intermediate (every rounding at its worst: a smooth function of the
pair, not the bit-dependent error one evaluation shows).  CPU only; the device-side sweeps and the adversarial
maximiser of tests/test_gpu_filter.py watch the OBSERVED error against the same bands.

  * fp32 filter form (ciede2000_fast<PREP>): band 1e-3 + 6e-5 v per pixel pair (engine.hip filter_band), two-sided;
    the bound stays below a quarter of it on every family -- the table's interpolation is most of it.
  * packed-fp16 pre-tier (ciede2000_h2): band launch.hpp kPreBandAbs + kPreBandRel v on the OVER-estimate (an under-estimate only
    delays an exit); the band IS this bound with a tenth of room, so the margin asserted here is that tenth.
    Round 4's band (0.29 + 8e-3 v, twice the worst OBSERVED pair) is a third of the bound: kept as an option, not the default.
"""
import os
import re
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import filter_error_bound as feb  # noqa: E402


def shipped_bands():
    """The constants as the sources hold them."""
    launch = open(os.path.join(ROOT, "mosaicmagnifique_amd", "csrc", "launch.hpp")).read()
    m = re.search(r"kPreBandAbs = ([0-9.e-]+)f, kPreBandRel = ([0-9.e-]+)f, kPreBandAbsObserved = ([0-9.e-]+)f, kPreBandRelObserved = ([0-9.e-]+)f", launch)
    engine = open(os.path.join(ROOT, "mosaicmagnifique_amd", "csrc", "engine.hip")).read()
    f = re.search(r"b\.px_abs = diff_type == MM_CIEDE2000 \? ([0-9.e-]+)f", engine)
    g = re.search(r"b\.rel = \(diff_type == MM_CIEDE2000 \? ([0-9.e-]+)f", engine)
    return tuple(float(v) for v in m.groups()), (float(f.group(1)), float(g.group(1)))


@pytest.fixture(scope="module")
def pairs():
    return feb.families(60000, np.random.default_rng(2024))


def test_the_tool_and_the_sources_name_the_same_bands():
    (pre_abs, pre_rel, _, _), (f_abs, f_rel) = shipped_bands()
    assert feb.BANDS["pretier"] == (pre_abs, pre_rel)
    assert feb.BANDS["filter"] == (f_abs, f_rel)


def test_reference_restatement_matches_the_known_answers():
    """ciede2000_ref is the yardstick of this file: the 34 Sharma pairs the reference's tests hold (tests/golden/colour_diff_kats.json)."""
    import json
    kats = json.load(open(os.path.join(ROOT, "tests", "golden", "colour_diff_kats.json")))
    rows = kats["cases"]["CIEDE2000"]["pairs"]
    a = np.array([r["first"] for r in rows], dtype=np.float64)
    b = np.array([r["second"] for r in rows], dtype=np.float64)
    want = np.array([r["difference"] for r in rows], dtype=np.float64)
    assert len(want) == 34
    assert np.abs(feb.ciede2000_ref(a, b) - want).max() <= 1e-4


def test_fp32_filter_band_holds_the_worst_case_bound_with_room(pairs):
    _, (f_abs, f_rel) = shipped_bands()
    worst = 0.0
    for name, (p, q) in pairs.items():
        true, _, _, need, handed = feb.evaluate(p, q, "filter")
        share = float((need / (f_abs + f_rel * true)).max())
        worst = max(worst, share)
        assert np.isfinite(need).all(), name
        assert share <= 0.25, (name, share)
    print("fp32 filter form: worst-case bound at %.3f of the band" % worst)


def test_fp16_pretier_band_is_the_worst_case_bound_with_a_tenth_of_room(pairs):
    (pre_abs, pre_rel, obs_abs, obs_rel), _ = shipped_bands()
    worst, worst_observed_band = 0.0, 0.0
    for name, (p, q) in pairs.items():
        true, _, _, need, seam = feb.evaluate(p, q, "pretier")
        assert np.isfinite(need).all(), name
        share = float((need / (pre_abs + pre_rel * true)).max())
        worst = max(worst, share)
        worst_observed_band = max(worst_observed_band, float((need / (obs_abs + obs_rel * true)).max()))
        assert share <= 0.92, (name, share)
    print("fp16 pre-tier: worst-case bound at %.3f of the shipped band, %.2f of round 4's observed band" % (worst, worst_observed_band))
    # what makes the observed band an option and not the default: the roundings, all at their worst, exceed it
    assert worst_observed_band > 1.5


def test_pairs_the_pretier_does_not_judge_count_zero(pairs):
    """Greys (a stored chroma below 1 / 16 on either side) and pairs within 1 / 128 of the opposite-hue seam contribute 0 to the
    pre-tier's sum -- an under-estimate, which only delays an exit."""
    p, q = pairs["u8 greys x u8 colours"]
    _, _, _, need, seam = feb.evaluate(p, q, "pretier")
    assert seam.all() and (need <= 0).all()
    p, q = pairs["contender-like"]
    assert feb.evaluate(p, q, "pretier")[4].mean() < 0.01


def test_euclidean_filter_band_holds_the_worst_case_bound():
    """RGB / CIE76: three differences, a sum of squares, one root -- the bound is a few ulp of the value, a fifth of the 5e-4 the engine trusts
    (engine.hip filter_band) at the largest value the validated box allows (|channel| <= 256: 887)."""
    rng = np.random.default_rng(3)
    p = rng.uniform(-256, 256, (200000, 3)).astype(np.float32).astype(np.float64)
    q = rng.uniform(-256, 256, (200000, 3)).astype(np.float32).astype(np.float64)
    p[:1000], q[:1000] = 256.0, -256.0                      # the corner of the box
    v, e = feb.euclid_bound(p, q)
    assert np.abs(v - np.sqrt(((p - q) ** 2).sum(-1))).max() < 1e-9
    print("Euclidean filter form: worst-case bound %.3g at a value of %.1f (band 5e-4)" % (e.max(), v[e.argmax()]))
    assert e.max() <= 0.5 * 5e-4


def test_algebraic_fp64_form_of_the_parallel_sums_is_good_to_a_few_1e_12(pairs):
    """colour_diff.hpp ciede2000_algebraic (what every parallel-order fp64 sum adds per pair since round 5) against CIEDE2000 in long
    double, every fp64 rounding and every library call (two ulp) at its worst: at most 2e-12 absolute per pair -- a cell's total is then
    good to 1e-13 relative, four orders inside the 1e-9 within which candidates are re-summed in the reference's order with the
    reference's own arithmetic (engine.hip kTieBand)."""
    worst = 0.0
    for name, (p, q) in pairs.items():
        true, _, _, need, handed = feb.evaluate(p[:20000], q[:20000], "algebraic")
        assert np.isfinite(need).all(), name
        worst = max(worst, float(need.max()))
        assert need.max() <= 2e-12, (name, float(need.max()))
    print("algebraic fp64 form: worst-case bound %.3g per pair" % worst)
