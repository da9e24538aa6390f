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


# ---------------------------------------------------------------------------------------------------------------------
# The bound AS THE DEVICE EVALUATES IT (filter.hip stats_bound: fp32, v_rsq / v_rcp / v_sqrt / v_exp / v_sin, block statistics that
# were themselves summed in fp32) against the same formula in exact arithmetic on exact statistics: with every rounding at its
# worst (the carried-bound arithmetic of tools/filter_error_bound.py; instruction accuracies as tests/ulp_sweep.hip measured
# them exhaustively) the device's value must not exceed the exact one -- that is what the formula's cushions (1e-2 of slack per
# block sum, 0.999 on the eigenvalue's leading term, 1.001 x + 1e-5 on the sine, 0.02 degrees on the arcs, 0.9999 overall) are for.
# Together with the tests above (exact formula <= true sum) this puts the device's bound below the truth.
# ---------------------------------------------------------------------------------------------------------------------
import os  # noqa: E402
import sys  # noqa: E402

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))


def device_bound_upper(cell, lib):
    """Upper end of what stats_bound can return for one 64-pixel block pair, every fp32 rounding at its worst."""
    import filter_error_bound as feb
    from filter_error_bound import EV, add, sub, mul, const, scale_pow2, monotone, absval
    feb.set_precision("single")
    try:
        u = 2.0 ** -24

        def stat_sum(px):                       # fp32 block sum: seven roundings along any path of the reduction
            return EV(px.sum(-1), 7 * u * np.abs(px).sum(-1))
        def stat_max(v):                        # a maximum of fp32 values each good to two ulp
            return EV(v.max(-1), 2 * u * np.abs(v).max(-1) + 1e-30)
        c_hc, c_hw = hue_arcs(cell)
        l_hc, l_hw = hue_arcs(lib)              # (the arcs carry + 0.02 degrees already; atan2f and the wraps: 1e-4 degrees)
        arc = lambda v: EV(v, 1e-4)
        c_sl, l_sl = stat_sum(cell[..., 0]), stat_sum(lib[..., 0])
        c_sa, l_sa = stat_sum(cell[..., 1]), stat_sum(lib[..., 1])
        c_sb, l_sb = stat_sum(cell[..., 2]), stat_sum(lib[..., 2])
        c_dev, l_dev = stat_max(np.abs(cell[..., 0] - 50)), stat_max(np.abs(lib[..., 0] - 50))
        c_cm, l_cm = stat_max(np.hypot(cell[..., 1], cell[..., 2])), stat_max(np.hypot(lib[..., 1], lib[..., 2]))
        x = scale_pow2(add(c_dev, l_dev), 0.5)
        x2 = mul(x, x)
        s_l = add(mul(mul(const(0.015), x2), monotone(add(EV(20.0), x2), lambda t: 1 / np.sqrt(t), floor=1.0)), EV(1.0))
        cbar = scale_pow2(add(c_cm, l_cm), 0.5)
        g_min = sub(EV(1.5), scale_pow2(feb.ratio7_single(mul(cbar, const(0.04))), 0.5))
        gc = mul(g_min, cbar)
        s_c = add(mul(const(0.045), gc), EV(1.0))
        # the least |h_bar' - 275| any pair can have: wraps, maxima and minima of sums -- 1-Lipschitz in every argument
        c1, w1, c2, w2 = arc(c_hc), arc(c_hw), arc(l_hc), arc(l_hw)
        delta = wrap180(c2.v - c1.v)
        mid = np.abs(wrap180(c1.v + 0.5 * delta))
        hw = 0.5 * (w1.v + w2.v)
        d0, d1 = np.maximum(0.0, mid - hw), np.maximum(0.0, (180.0 - mid) - hw)
        off_v = np.where(np.abs(delta) + w1.v + w2.v >= 180.0, np.minimum(d0, d1), d0)
        off = EV(off_v, 4e-4 + 8 * u * 360.0)
        # e_max = exp2(-off^2 log2(e) / 625): decreasing in off (interval ends), v_exp_f32 one ulp (measured 0.78)
        ex = lambda t: np.exp(-(np.maximum(t, 0.0) / 25.0) ** 2)
        e_v = ex(off.v)
        e_max = EV(e_v, np.maximum(np.abs(ex(off.v - off.e) - e_v), np.abs(ex(off.v + off.e) - e_v)) + 4 * u * e_v)
        # sin(60 deg x e_max) through v_sin_f32 (absolute error 1.8 x 2^-24 measured, 4 charged), then x 1.001 + 1e-5
        sn = lambda t: np.sin(np.radians(60.0) * np.clip(t, 0.0, 1.0))
        s_v = sn(e_max.v)
        sin_2dt = add(mul(EV(s_v, np.maximum(np.abs(sn(e_max.v - e_max.e) - s_v), np.abs(sn(e_max.v + e_max.e) - s_v)) + 4 * u), const(1.001)), const(1e-5))
        half_rt = mul(feb.ratio7_single(mul(gc, const(0.04))), sin_2dt)
        slack = EV(1e-2)
        def pos(a):                             # max(., 0) of an interval: a difference that stays below its slack is exactly 0 on the device
            lo, hi = np.maximum(a.v - a.e, 0.0), np.maximum(a.v + a.e, 0.0)
            return EV(0.5 * (lo + hi), 0.5 * (hi - lo))
        t_l = mul(pos(sub(absval(sub(c_sl, l_sl)), slack)), monotone(s_l, lambda t: 1 / t, floor=1.0))
        da = mul(pos(sub(absval(sub(c_sa, l_sa)), slack)), g_min)
        db = pos(sub(absval(sub(c_sb, l_sb)), slack))
        inv_c = monotone(s_c, lambda t: 1 / t, floor=1.0)
        inv_h = monotone(add(mul(const(0.015 * 1.5726), gc), EV(1.0)), lambda t: 1 / t, floor=1.0)
        na, nc = mul(inv_c, inv_c), mul(inv_h, inv_h)
        nb = mul(mul(half_rt, inv_c), inv_h)
        half_diff = scale_pow2(sub(na, nc), 0.5)
        root = monotone(add(mul(half_diff, half_diff), mul(nb, nb)), np.sqrt)
        lam = pos(sub(mul(const(0.999), scale_pow2(add(na, nc), 0.5)), root))
        total = add(mul(t_l, t_l), mul(lam, add(mul(da, da), mul(db, db))))
        out = mul(const(0.9999), monotone(total, np.sqrt))
        return out.v + out.e
    finally:
        feb.set_precision("half")


def exact_block_bound(cell, lib):
    """The same formula in float64 on exact statistics, without any cushion (what the tests above hold below the true sum,
    up to the 1.0001 they put on R_T)."""
    dev = 0.5 * (np.abs(cell[..., 0] - 50).max(-1) + np.abs(lib[..., 0] - 50).max(-1))
    s_l = 1.0 + 0.015 * dev ** 2 / np.sqrt(20.0 + dev ** 2)
    cbar = 0.5 * (np.hypot(cell[..., 1], cell[..., 2]).max(-1) + np.hypot(lib[..., 1], lib[..., 2]).max(-1))
    g_min = 1.5 - 0.5 * sqrt_ratio7(cbar)
    gc = g_min * cbar
    s_c, s_h = 1.0 + 0.045 * gc, 1.0 + 0.015 * 1.5726 * gc
    half_rt = sqrt_ratio7(gc) * np.sin(np.radians(60.0) * rotation_bound(cell, lib))
    a, c, b = 1.0 / s_c ** 2, 1.0 / s_h ** 2, half_rt / (s_c * s_h)
    lam = np.maximum(0.5 * (a + c) - np.sqrt(0.25 * (a - c) ** 2 + b ** 2), 0.0)
    d = np.abs(cell.sum(-2) - lib.sum(-2))
    return np.sqrt((d[..., 0] / s_l) ** 2 + lam * ((g_min * d[..., 1]) ** 2 + d[..., 2] ** 2))


@pytest.mark.parametrize("kind", ["uniform", "patches", "close-patches", "blue", "blue-chroma-shift", "hue-pairs", "hue-pairs-wide",
                                  "mean-at-275", "seam-95", "opposite", "near-grey", "dark-vs-bright"])
def test_the_bound_as_the_device_evaluates_it_stays_below_the_exact_formula(kind):
    rng = np.random.default_rng(31)
    cell, lib = groups(rng, kind, 20000, n=64)                       # one block = 64 positions
    cell, lib = cell.astype(np.float32).astype(np.float64), lib.astype(np.float32).astype(np.float64)
    upper, exact = device_bound_upper(cell, lib), exact_block_bound(cell, lib)
    assert np.isfinite(upper).all()
    worst = (upper - exact).max()
    print("%s: device bound (worst roundings) - exact bound: max %.3g, min margin %.3g of the exact bound" %
          (kind, worst, ((exact - upper) / np.maximum(exact, 1e-9)).min()))
    assert (upper <= exact + 1e-9).all(), (kind, float(worst))


def test_block_sums_of_the_euclidean_forms_stay_inside_their_slack():
    """filter_lds_kernel's block bound (RGB / CIE76: the plain triangle inequality, sum ||d|| >= ||sum d||) takes 2e-2 off every
    |sum_cell - sum_lib| for the same two fp32 block sums, of values up to 256 (the box the Euclidean filter forms are validated on):
    worst error of the device's summation order, twice, plus the difference's rounding."""
    rng = np.random.default_rng(9)
    n = 200000
    x = rng.uniform(0, 256, (n, 16, 4)).astype(np.float32)
    x[: n // 2] = (256.0 - rng.uniform(0, 1e-3, (n // 2, 16, 4))).astype(np.float32)
    lane = x[:, :, 0]
    for k in range(1, 4):
        lane = (lane + x[:, :, k]).astype(np.float32)
    for off in (8, 4, 2, 1):
        lane = (lane + lane[:, np.arange(16) ^ off]).astype(np.float32)
    err = np.abs(lane[:, 0].astype(np.float64) - x.astype(np.float64).sum((1, 2))).max()
    diff_rounding = 2.0 ** -24 * 64 * 256.0
    print("Euclidean block sums: worst fp32 error %.3g (x 2 images + %.3g) against a slack of 2e-2" % (err, diff_rounding))
    assert 2 * err + diff_rounding <= 2e-2
