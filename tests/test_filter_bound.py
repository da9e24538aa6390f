"""The lower bound behind the filter kernel's early exit (csrc/filter.hip, filter_kernel):

    sum over a 256-pixel group of dE00(cell pixel, library pixel)
        >= sqrt( (|sum dL|)^2 / S_L^2 + rho ((|sum da|)^2 + (|sum db|)^2) / S_C^2 )

with S_L, S_C, rho the worst constants the group's lightness range and largest chroma allow.
A candidate is dropped when (partial sum + bound of the unvisited groups) passes the cell's
limit, so the bound must NEVER exceed the true sum: checked here against the oracle's
CIEDE2000 on random and adversarial groups (the numpy restatement below is the kernel's
formula in float64; the kernel itself is held to the unpruned path in test_gpu_filter.py)."""
import numpy as np
import pytest


def sqrt_ratio7(x):
    u = (x / 25.0) ** 7
    return np.sqrt(u / (u + 1.0))


def wrap180(v):
    """Angle in degrees wrapped into [-180, 180] (round-half-even, like v_rndne_f32)."""
    return v - 360.0 * np.rint(v / 360.0)


def hue_arcs(px):
    """Per block: the arc of the hue circle that holds every pixel's hue, as (centre, half-width) in
    degrees, the centre given as its offset from 275 deg in [-180, 180] -- the kernel's statistics 5 and 6
    (filter.hip group_sums_kernel).  A pixel's hue is that of (g a, b) for every g = 1 + G in [1, 1.5]
    (ColourDifference.cpp:50-53,61-83): both ends are taken.  Deviations are measured from the direction
    of the block's summed (a, b), so no fixed seam cuts through a block; the whole circle (half-width 180)
    when that sum vanishes, when a pixel's chroma is (nearly) zero -- its hue is arbitrary and the
    reference's mean-hue rule changes (:108-109) -- or when a pixel lies more than 170 deg off the sum."""
    a, b = px[..., 1], px[..., 2]
    sa, sb = a.sum(-1, keepdims=True), b.sum(-1, keepdims=True)
    ref = np.degrees(np.arctan2(sb, sa))
    d1 = wrap180(np.degrees(np.arctan2(b, a)) - ref)
    d2 = wrap180(np.degrees(np.arctan2(b, 1.5 * a)) - ref)
    lo, hi = np.minimum(d1, d2).min(-1), np.maximum(d1, d2).max(-1)
    bad = (np.hypot(a, b).min(-1) < 1e-3) | (np.hypot(sa, sb)[..., 0] < 1e-3) | (np.maximum(np.abs(d1), np.abs(d2)).max(-1) > 170.0)
    centre = wrap180(ref[..., 0] + 0.5 * (lo + hi) - 275.0)
    half = 0.5 * (hi - lo) + 0.02
    return np.where(bad, 0.0, centre), np.where(bad, 180.0, half)


def rotation_bound(cell, lib):
    """Upper bound of exp(-((h_bar' - 275) / 25)^2) over all pixel pairs of the two blocks.  A pair's mean
    hue is the midpoint of the shorter arc between its two hues (ColourDifference.cpp:107-122); with the
    blocks' arcs (c1, w1), (c2, w2) and D = wrap(c2 - c1) that midpoint lies within (w1 + w2) / 2 of
    c1 + D / 2 -- or of the opposite point when some pair can be further than 180 deg apart
    (|D| + w1 + w2 >= 180).  The reference does not wrap h_bar' - 275; the circular distance used here
    is never larger, so the exponent is never more negative than the true one."""
    c1, w1 = hue_arcs(cell)
    c2, w2 = hue_arcs(lib)
    delta = wrap180(c2 - c1)
    mid = np.abs(wrap180(c1 + 0.5 * delta))
    hw = 0.5 * (w1 + w2)
    d0 = np.maximum(0.0, mid - hw)
    d1 = np.maximum(0.0, (180.0 - mid) - hw)
    d = np.where(np.abs(delta) + w1 + w2 >= 180.0, np.minimum(d0, d1), d0)
    return np.exp(-(d / 25.0) ** 2)


def group_bound(cell, lib, hue_aware=True):
    """cell, lib: [..., n, 3] Lab pixels of one group -> lower bound of sum_p dE00."""
    dev = 0.5 * (np.abs(cell[..., 0] - 50).max(-1) + np.abs(lib[..., 0] - 50).max(-1))
    s_l = 1.0 + 0.015 * dev ** 2 / np.sqrt(20.0 + dev ** 2)
    cbar = 0.5 * (np.hypot(cell[..., 1], cell[..., 2]).max(-1) + np.hypot(lib[..., 1], lib[..., 2]).max(-1))
    g_min = 1.5 - 0.5 * sqrt_ratio7(cbar)        # 1 + G of every pair is at least this: G falls as the pair's mean chroma grows (:50-53)
    gc = g_min * cbar
    s_c = 1.0 + 0.045 * gc
    e_max = rotation_bound(cell, lib) if hue_aware else 1.0
    r_t = 2.0 * sqrt_ratio7(gc) * np.sin(np.radians(60.0) * e_max) * 1.0001          # >= |R_T|, R_C = 2 sqrt_ratio7
    # tC^2 + tH^2 + R_T tC tH = (dC', dH') N (dC', dH')^T with N = [[1/S_C^2, R_T/(2 S_C S_H)], [., 1/S_H^2]]: its smallest
    # eigenvalue falls as S_C, S_H or |R_T| grow, so the worst constants of the block give a floor; S_H <= 1 + 0.015 * 1.5726 g
    s_h = 1.0 + 0.015 * 1.5726 * gc
    a, c, b = 1.0 / s_c ** 2, 1.0 / s_h ** 2, r_t / (2.0 * s_c * s_h)
    lam = np.maximum(0.5 * (a + c) - np.sqrt(0.25 * (a - c) ** 2 + b ** 2), 0.0)
    d = np.abs(cell.sum(-2) - lib.sum(-2))
    # dC'^2 + dH'^2 = |(a1', b1) - (a2', b2)|^2 = ((1 + G) da)^2 + db^2 >= (g_min da)^2 + db^2
    return np.sqrt((d[..., 0] / s_l) ** 2 + lam * ((g_min * d[..., 1]) ** 2 + d[..., 2] ** 2))


def groups(rng, kind, n_groups, n=256):
    lo, hi = np.array([0, -128, -128.0]), np.array([100, 127, 127.0])
    if kind == "uniform":
        return rng.uniform(lo, hi, (n_groups, n, 3)), rng.uniform(lo, hi, (n_groups, n, 3))
    if kind == "patches":                 # what photos and the synthetic library look like: a base colour + noise
        base = lambda: rng.uniform(lo, hi, (n_groups, 1, 3))
        return base() + rng.normal(0, 6, (n_groups, n, 3)), base() + rng.normal(0, 6, (n_groups, n, 3))
    if kind == "close-patches":           # contenders: nearly the same base colour
        b = rng.uniform(lo, hi, (n_groups, 1, 3))
        return b + rng.normal(0, 4, (n_groups, n, 3)), b + rng.normal(0, 1.5, (n_groups, 1, 3)) + rng.normal(0, 4, (n_groups, n, 3))
    if kind == "blue":                    # hue ~275 deg, high chroma: the rotation term R_T at its largest
        def blue():
            th = np.deg2rad(rng.uniform(255, 295, (n_groups, 1)) + rng.normal(0, 4, (n_groups, n)))
            r = rng.uniform(30, 120, (n_groups, 1)) + rng.normal(0, 5, (n_groups, n))
            return np.stack([rng.uniform(10, 90, (n_groups, 1)) + rng.normal(0, 4, (n_groups, n)), r * np.cos(th), r * np.sin(th)], -1)
        return blue(), blue()
    if kind == "blue-chroma-shift":       # same hue and lightness, different chroma: only tC (and R_T) speak
        th = np.deg2rad(275.0)
        def ray(r):
            return np.stack([np.full_like(r, 50.0), r * np.cos(th), r * np.sin(th)], -1)
        r1 = rng.uniform(20, 100, (n_groups, 1)) + rng.normal(0, 1, (n_groups, n))
        return ray(r1), ray(r1 + rng.uniform(-15, 15, (n_groups, 1)))
    if kind in ("hue-pairs", "hue-pairs-wide"):       # two narrow hue clusters at every relative position, incl. means at 275 and opposite hues
        spread = 3.0 if kind == "hue-pairs" else 25.0
        def cluster():
            th = np.deg2rad(rng.uniform(0, 360, (n_groups, 1)) + rng.uniform(-spread, spread, (n_groups, n)))
            r = rng.uniform(15, 110, (n_groups, 1)) + rng.normal(0, 3, (n_groups, n))
            return np.stack([rng.uniform(20, 80, (n_groups, 1)) + rng.normal(0, 2, (n_groups, n)), r * np.cos(th), r * np.sin(th)], -1)
        return cluster(), cluster()
    if kind == "mean-at-275":             # hue clusters placed symmetrically about 275 deg (mean hue exactly where R_T peaks)
        off = rng.uniform(0, 175, (n_groups, 1))
        def side(sign):
            th = np.deg2rad(275.0 + sign * off + rng.uniform(-2, 2, (n_groups, n)))
            r = rng.uniform(30, 110, (n_groups, 1)) + rng.normal(0, 2, (n_groups, n))
            return np.stack([rng.uniform(30, 70, (n_groups, 1)) + rng.normal(0, 2, (n_groups, n)), r * np.cos(th), r * np.sin(th)], -1)
        return side(+1), side(-1)
    if kind in ("seam-95", "opposite"):   # hues around 95 deg (opposite 275: where a fixed seam would cut the block);
        def cluster(centre, spread):      # clusters (nearly) opposite each other: the mean hue flips by 180 deg inside the block pair
            th = np.deg2rad(centre + rng.uniform(-spread, spread, (n_groups, n)))
            r = rng.uniform(15, 110, (n_groups, 1)) + rng.normal(0, 3, (n_groups, n))
            return np.stack([rng.uniform(20, 80, (n_groups, 1)) + rng.normal(0, 2, (n_groups, n)), r * np.cos(th), r * np.sin(th)], -1)
        if kind == "seam-95":
            return cluster(95.0 + rng.uniform(-30, 30, (n_groups, 1)), 20.0), cluster(95.0 + rng.uniform(-30, 30, (n_groups, 1)), 20.0)
        base = rng.uniform(0, 360, (n_groups, 1))
        return cluster(base, 4.0), cluster(base + 180.0 + rng.uniform(-6, 6, (n_groups, 1)), 4.0)
    if kind == "grey-pixel-inside":       # colourful blocks with one exactly grey pixel each (hue undefined there)
        a, b = groups(rng, "hue-pairs", n_groups, n)
        a[:, 0, 1:] = 0.0; b[:, 1, 1:] = 0.0
        return a, b
    if kind == "near-grey":
        a, b = rng.uniform(lo, hi, (n_groups, n, 3)), rng.uniform(lo, hi, (n_groups, n, 3))
        a[..., 1:] *= 0.02; b[..., 1:] *= 0.02
        return a, b
    if kind == "dark-vs-bright":
        a, b = rng.uniform(lo, hi, (n_groups, n, 3)), rng.uniform(lo, hi, (n_groups, n, 3))
        a[..., 0] = rng.uniform(0, 15, (n_groups, n)); b[..., 0] = rng.uniform(85, 100, (n_groups, n))
        return a, b
    raise ValueError(kind)


@pytest.mark.parametrize("n", [256, 64])
@pytest.mark.parametrize("kind", ["uniform", "patches", "close-patches", "blue", "blue-chroma-shift", "near-grey", "dark-vs-bright",
                                  "hue-pairs", "hue-pairs-wide", "mean-at-275", "grey-pixel-inside", "seam-95", "opposite"])
def test_group_bound_never_exceeds_the_true_sum(oracle, kind, n):
    rng = np.random.default_rng(sum(map(ord, kind)) + n)
    cell, lib = groups(rng, kind, 1500, n)
    cell, lib = cell.astype(np.float32), lib.astype(np.float32)        # what the engine holds
    true = oracle.colour_diff_batch(2, cell.reshape(-1, 3), lib.reshape(-1, 3)).reshape(cell.shape[:2]).sum(1)
    bound = group_bound(cell.astype(np.float64), lib.astype(np.float64))
    assert (bound <= true * (1 + 1e-9) + 1e-9).all(), (kind, (bound / np.maximum(true, 1e-30)).max())
    if kind in ("patches", "dark-vs-bright"):
        assert np.median(bound / true) > 0.25           # and it is worth having
    if kind == "hue-pairs":                             # the hue-aware rotation bound is what makes it tight here
        assert np.median(bound / true) > 1.3 * np.median(group_bound(cell.astype(np.float64), lib.astype(np.float64), hue_aware=False) / true)


def test_the_constants_are_worst_cases(oracle):
    """g(C) = (1.5 - 0.5 sqrt(C^7 / (C^7 + 25^7))) C grows with C (so the largest chroma of the two
    groups bounds every pair's C'), and T stays below 1.93 (so S_H <= S_C)."""
    c = np.linspace(0, 200, 200001)
    g = (1.5 - 0.5 * sqrt_ratio7(c)) * c
    assert (np.diff(g) > 0).all()
    h = np.deg2rad(np.linspace(0, 360, 360001))
    t = 1 - 0.17 * np.cos(h - np.deg2rad(30)) + 0.24 * np.cos(2 * h) + 0.32 * np.cos(3 * h + np.deg2rad(6)) - 0.20 * np.cos(4 * h - np.deg2rad(63))
    assert 0 < t.min() and t.max() < 1.5726 and 0.015 * t.max() < 0.045


@pytest.mark.parametrize("type_id,kind", [(0, "uniform"), (0, "patches"), (1, "close-patches"), (1, "blue")])
def test_euclid_block_bound_never_exceeds_the_true_sum(oracle, type_id, kind):
    """RGB-Euclidean / CIE76: sum_block |x_p - y_p| >= |sum_block x_p - sum_block y_p| (triangle
    inequality of the Euclidean norm), the bound of filter_lds_kernel's early exit -- no constants."""
    rng = np.random.default_rng(7 + sum(map(ord, kind)))
    cell, lib = groups(rng, kind, 2000, n=64)
    if type_id == 0:                                       # what RGB inputs look like: integer-valued 0..255
        cell = np.clip(np.rint(cell * 2.0 + 60), 0, 255); lib = np.clip(np.rint(lib * 2.0 + 60), 0, 255)
    cell, lib = cell.astype(np.float32), lib.astype(np.float32)
    true = oracle.colour_diff_batch(type_id, cell.reshape(-1, 3), lib.reshape(-1, 3)).reshape(cell.shape[:2]).sum(1)
    d = np.abs(cell.astype(np.float64).sum(1) - lib.astype(np.float64).sum(1))
    bound = np.sqrt((d ** 2).sum(1))
    assert (bound <= true * (1 + 1e-12) + 1e-9).all()
    if kind == "patches":
        assert np.median(bound / true) > 0.5


def test_block_sums_in_fp32_stay_inside_the_slack_the_bound_grants_them():
    """`stats_bound` subtracts 1e-2 from |sum_cell - sum_lib| of every channel before it uses it: what the fp32 rounding of the two
    64-term block sums (filter.hip group_sums_kernel: four values per lane added in turn, then four xor-shuffle stages over the
    block's sixteen lanes -- seven roundings along any path) and of their difference can hide.  Values at the edge of the validated
    box (|a|, |b| <= 150): the worst |fp32 sum - exact sum| over 2e5 blocks, twice (two images), plus the difference's own rounding,
    stays below the slack."""
    rng = np.random.default_rng(8)
    n = 200000
    x = rng.uniform(-150, 150, (n, 16, 4)).astype(np.float32)
    x[: n // 4] = np.abs(x[: n // 4])                          # same-sign blocks: the largest partial sums
    x[n // 4: n // 2] = (150.0 - rng.uniform(0, 1e-3, (n // 4, 16, 4))).astype(np.float32)      # everything at the edge
    lane = x[:, :, 0]
    for k in range(1, 4):
        lane = (lane + x[:, :, k]).astype(np.float32)          # s += xs[k], in turn
    for off in (8, 4, 2, 1):                                   # s += shfl_xor(s, off): every lane ends with the block's sum
        lane = (lane + lane[:, np.arange(16) ^ off]).astype(np.float32)
    exact = x.astype(np.float64).sum((1, 2))
    err = np.abs(lane[:, 0].astype(np.float64) - exact).max()
    diff_rounding = 2.0 ** -24 * 2 * 64 * 150.0                # |sum_cell - sum_lib| <= 19 200, one rounding
    print("block sums: worst fp32 error %.3g (x 2 images + %.3g for the difference) against a slack of 1e-2" % (err, diff_rounding))
    assert 2 * err + diff_rounding <= 1e-2
    assert err <= 7 * 2.0 ** -24 * 64 * 150.0                  # the first-order bound quoted in DESIGN.md section 2
