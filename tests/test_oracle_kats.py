"""The oracle against every known-answer vector the reference's tests hold for the
path (tests/golden/colour_diff_kats.json <- test/tst_ColourDifference.h:26-35,
:49-56, :73-108), at the reference's own tolerances."""
import json
import os

import numpy as np
import pytest

KATS = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "colour_diff_kats.json")))


def test_fixture_is_complete():
    assert len(KATS["cases"]["RGBEuclidean"]["pairs"]) == 8
    assert len(KATS["cases"]["CIE76"]["pairs"]) == 6
    assert len(KATS["cases"]["CIEDE2000"]["pairs"]) == 34       # Sharma et al. 2005


@pytest.mark.parametrize("name", ["RGBEuclidean", "CIE76", "CIEDE2000"])
def test_oracle_matches_reference_kats(oracle, name):
    case = KATS["cases"][name]
    for pair in case["pairs"]:
        got = oracle.colour_diff(case["type"], pair["first"], pair["second"])
        assert abs(got - pair["difference"]) <= case["tolerance"], pair


def test_batch_promotes_f32_like_vec3f(oracle):
    rng = np.random.default_rng(1)
    a = rng.uniform(-100, 100, (257, 3)).astype(np.float32)
    b = rng.uniform(-100, 100, (257, 3)).astype(np.float32)
    for t in (0, 2):
        got = oracle.colour_diff_batch(t, a, b)
        for i in (0, 100, 256):
            want = oracle.colour_diff(t, [float(v) for v in a[i]], [float(v) for v in b[i]])
            assert got[i] == want


def test_ciede2000_zero_chroma_and_identity(oracle):
    assert oracle.colour_diff(2, [50, 0, 0], [50, 0, 0]) == 0.0
    assert oracle.colour_diff(2, [0, 0, 0], [0, 0, 0]) == 0.0
    d = oracle.colour_diff(2, [0, 0, 0], [63.5, 12.0, -40.0])
    assert np.isfinite(d) and d > 0


def test_unknown_type_is_nan(oracle):
    assert np.isnan(oracle.colour_diff(3, [0, 0, 0], [1, 1, 1]))
