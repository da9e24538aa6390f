"""Worst-case error bounds of the engine's approximate CIEDE2000 forms:

    --form filter     the fp32 filter form       (csrc/colour_diff_fast.hpp ciede2000_fast<PREP>; band: engine.hip filter_band)
    --form pretier    the packed-fp16 pre-tier   (csrc/colour_diff_fast.hpp ciede2000_h2;        band: launch.hpp kPreBandAbs / kPreBandRel)
    --form algebraic  the fp64 parallel sums     (csrc/colour_diff.hpp ciede2000_algebraic;      against CIEDE2000 in long double)

What the bands had behind them until round 5 were OBSERVED errors: device sweeps and an adversarial maximiser.  An observed error
is one realisation of the roundings at one input -- jagged in the input's bits, which is how round 4's sweeps stepped over what the
maximiser then found in the fp16 form.  This tool bounds, for a given input pair, EVERY realisation: it evaluates the form's own
sequence of operations in float64 (long double for the fp64 form) and carries, beside each intermediate value, a bound on how far
the device's computation can be from it -- the running error analysis of Higham, "Accuracy and Stability of Numerical
Algorithms", sec. 3.3, with interval endpoints instead of derivatives wherever a function is monotone (rsq, rcp, sqrt: no
first-order truncation), every rounding charged at its worst (u = 2^-11 / 2^-24 / 2^-53 per operation, one ulp per transcendental --
tests/ulp_sweep.hip measures <= 0.92 ulp exhaustively --, subnormal halfs at their absolute step), constants charged with their own
rounding, the hue table's index moved by the error of the mean-hue direction and priced at the table's steepest step within reach,
a select whose condition the error can flip evaluated on both arms, and the places where a computed quantity enters numerator and
denominator alike (1 + G in a1' and a2'; the 7th-power ratio; S_L, S_C, S_H against the single reciprocal; the quadratic form
dC'^2 + dH'^2 + R_T dC' dH') treated as the device computes them instead of being charged twice.  The result

    error(x)  <=  |m(x)| + eps(x),     m = (the form in exact arithmetic) - (CIEDE2000),   eps = the carried bound

(for the pre-tier the signed OVER-estimate m + eps: an under-estimate only delays an early exit) is a SMOOTH function of the input
(no dependence on mantissa bits), so a sample that is dense in the form's weak zones says something about the points between its
nodes, which a sweep of observed errors does not.  It is still a sampled statement -- the input space is six-dimensional -- not an
interval proof over boxes: DESIGN.md section 2 says so.

    python3 tools/filter_error_bound.py --form filter|pretier|algebraic [--pairs N] [--abs A --rel R]

prints the worst bound per family as a share of the band, and the smallest (abs, rel) frontier that holds all of them.
tests/test_filter_error_bound.py holds the shipped bands above the bounds on a fixed subset (CPU only, seconds);
tests/test_filter_bound.py uses the same arithmetic on the block bound of the early exit (stats_bound).
"""
import argparse
import math
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
U = 2.0 ** -11           # round-to-nearest half: relative
ULP = 2.0 ** -10         # v_rsq_f16 / v_rcp_f16 / v_sqrt_f16: one ulp (ISA guide), relative
ETA = 2.0 ** -25         # half an ulp of a subnormal half: absolute
TINY = 2.0 ** -14        # smallest normal half
HMAX = 65504.0


def set_precision(kind):
    """'half': the packed-fp16 pre-tier (the defaults above); 'single': the fp32 filter form -- u = 2^-24, v_rsq / v_rcp / v_sqrt_f32
    one ulp, denormals flushed (what lies below 2^-126 counts as an error of its own size, at most 2^-126)."""
    global U, ULP, ETA, TINY, HMAX, DT
    DT = np.float64
    if kind == "half":
        U, ULP, ETA, TINY, HMAX = 2.0 ** -11, 2.0 ** -10, 2.0 ** -25, 2.0 ** -14, 65504.0
    elif kind == "single":
        U, ULP, ETA, TINY, HMAX = 2.0 ** -24, 2.0 ** -23, 2.0 ** -126, 2.0 ** -126, 3.0e38
    else:       # 'double': the algebraic fp64 form of the parallel sums; the "exact" values are then carried in long double (64-bit mantissa)
        U, ULP, ETA, TINY, HMAX = 2.0 ** -53, 2.0 ** -52, 0.0, 0.0, 1.0e300
        DT = np.longdouble


DT = np.float64
N_LUT = 2048
TINY_CHROMA = 2.0 ** -6   # in the stored units (C / 4): a chroma of 1 / 16


# --------------------------------------------------------------------------- CIEDE2000 in float64 (ColourDifference.cpp:42-158)
def ciede2000_ref(p, q):
    """p, q: [..., 3] Lab.  The reference's own sequence (radians, the three hue branches)."""
    L1, a1, b1 = p[..., 0], p[..., 1], p[..., 2]
    L2, a2, b2 = q[..., 0], q[..., 1], q[..., 2]
    k25 = 25.0 ** 7
    c1, c2 = np.hypot(a1, b1), np.hypot(a2, b2)
    cb = 0.5 * (c1 + c2)
    cb7 = cb ** 7
    g = 0.5 * (1.0 - np.sqrt(cb7 / (cb7 + k25)))
    a1p, a2p = (1.0 + g) * a1, (1.0 + g) * a2
    c1p, c2p = np.hypot(a1p, b1), np.hypot(a2p, b2)
    two_pi = 2.0 * math.pi

    def hue(b, ap):
        h = np.arctan2(b, ap)
        h = np.where(h < 0, h + two_pi, h)
        return np.where((b == 0) & (ap == 0), 0.0, h)
    h1, h2 = hue(b1, a1p), hue(b2, a2p)
    dl, dc = L2 - L1, c2p - c1p
    dh = h2 - h1
    zero = (c1p * c2p) == 0
    dh = np.where(zero, 0.0, np.where(dh > math.pi, dh - two_pi, np.where(dh < -math.pi, dh + two_pi, dh)))
    dH = 2.0 * np.sqrt(c1p * c2p) * np.sin(dh / 2.0)
    lb, cbp = 0.5 * (L1 + L2), 0.5 * (c1p + c2p)
    hs = h1 + h2
    hb = np.where(zero, hs, np.where(np.abs(h1 - h2) <= math.pi, hs / 2.0,
                                     np.where(hs < two_pi, (hs + two_pi) / 2.0, (hs - two_pi) / 2.0)))
    rad = math.radians
    T = 1.0 - 0.17 * np.cos(hb - rad(30)) + 0.24 * np.cos(2 * hb) + 0.32 * np.cos(3 * hb + rad(6)) - 0.20 * np.cos(4 * hb - rad(63))
    dth = rad(30) * np.exp(-(((hb - rad(275)) / rad(25)) ** 2))
    cbp7 = cbp ** 7
    rc = 2.0 * np.sqrt(cbp7 / (cbp7 + k25))
    sl = 1.0 + 0.015 * (lb - 50) ** 2 / np.sqrt(20 + (lb - 50) ** 2)
    sc = 1.0 + 0.045 * cbp
    sh = 1.0 + 0.015 * cbp * T
    rt = -np.sin(2 * dth) * rc
    tl, tc, th = dl / sl, dc / sc, dH / sh
    return np.sqrt(np.maximum(tl * tl + tc * tc + th * th + rt * tc * th, 0.0))


def hue_lut():
    """engine.hip build_hue_lut: entries {0.0075 T, -2 sin(2 d_theta)} over the diamond angle, as the f32 values the device holds."""
    i = np.arange(N_LUT + 2) % N_LUT
    t = 4.0 * i / N_LUT
    upper = t <= 2.0
    q = np.where(upper, 1.0 - t, t - 3.0)
    hc, hs = q, np.where(upper, 1.0, -1.0) * (1.0 - np.abs(q))
    h = np.arctan2(hs, hc)
    h = np.where(h < 0, h + 2 * math.pi, h)
    rad = math.radians
    T = 1.0 - 0.17 * np.cos(h - rad(30)) + 0.24 * np.cos(2 * h) + 0.32 * np.cos(3 * h + rad(6)) - 0.20 * np.cos(4 * h - rad(63))
    z = (h - rad(275)) / rad(25)
    G = -2.0 * np.sin(2 * rad(30) * np.exp(-z * z))
    return (0.0075 * T).astype(np.float32).astype(np.float64), G.astype(np.float32).astype(np.float64)


LUT_T, LUT_G = hue_lut()


def window_slopes(lut):
    """level k, entry i: the largest step between neighbouring entries within 2^k entries of i (the circle closes): what an
    index error of one costs at most in that neighbourhood."""
    step = np.abs(np.diff(lut[:N_LUT + 1]))                       # step[i] = |lut[i + 1] - lut[i]|, i < N
    levels = [np.maximum(step, np.roll(step, 1))]
    for k in range(1, 11):
        prev, h = levels[-1], 2 ** (k - 1)
        levels.append(np.maximum(prev, np.maximum(np.roll(prev, h), np.roll(prev, -h))))
    return np.stack(levels)


SLOPES_T, SLOPES_G = window_slopes(LUT_T), window_slopes(LUT_G)


# --------------------------------------------------------------------------- values with a carried error bound
class EV:
    """v: the value in exact arithmetic (float64); e: bound on |half-precision computation - v|."""
    __slots__ = ("v", "e")

    def __init__(self, v, e=None):
        self.v = np.asarray(v, dtype=DT)
        self.e = np.zeros_like(self.v) if e is None else np.asarray(e, dtype=DT)

    def mag(self):
        return np.abs(self.v) + self.e


def rounded(v, e, rel=None):
    """One rounding of a computed result whose exact counterpart is v with incoming error e: relative above the subnormal
    halfs; half a subnormal step below 2^-14; and what lies below half a step is rounded to zero -- an error of its own size."""
    rel = U if rel is None else rel
    mag = np.abs(v) + e
    return EV(v, e + np.where(mag < TINY, np.minimum(mag, ETA), rel * mag))


def const(c):
    """A constant of the source as the half (float) the compiler stores."""
    h = float(np.float16(c)) if U > 1e-5 else (float(np.float32(c)) if U > 1e-10 else float(c))
    return EV(c, abs(h - c))


def add(a, b, sign=1.0):
    return rounded(a.v + sign * b.v, a.e + b.e)


def sub(a, b):
    return add(a, b, -1.0)


def mul(a, b):
    return rounded(a.v * b.v, np.abs(a.v) * b.e + np.abs(b.v) * a.e + a.e * b.e)


def scale_pow2(a, s):
    """Multiplication by an exact power of two: no rounding above the subnormals."""
    mag = (np.abs(a.v) + a.e) * abs(s)
    return EV(a.v * s, a.e * abs(s) + np.where(mag < TINY, np.minimum(mag, ETA), 0.0))


def fma_like(a, b, c, sign=1.0):
    """a * b + sign * c as the compiler may or may not contract it: two roundings bound one."""
    return add(mul(a, b), c, sign)


def monotone(a, f, rel=None, floor=0.0):
    """f monotone on [v - e, v + e] (clipped at `floor`): the interval's endpoints bound the propagated error."""
    rel = ULP if rel is None else rel
    lo, hi = np.maximum(a.v - a.e, floor), a.v + a.e
    with np.errstate(divide="ignore", invalid="ignore", over="ignore"):
        fv, flo, fhi = f(np.maximum(a.v, floor)), f(lo), f(hi)
    prop = np.maximum(np.abs(flo - fv), np.abs(fhi - fv))
    prop = np.where(np.isfinite(prop), prop, np.inf)
    return rounded(fv, prop, rel)


def vmax(a, b):
    """max(a, b) as an interval: [max of the lower ends, max of the upper ends] -- the floor b holds the result up whatever a's
    error does (what the chroma floor of ciede2000_h2 is for); returned as centre and half-width."""
    lo = np.maximum(a.v - a.e, b.v - b.e)
    hi = np.maximum(a.v + a.e, b.v + b.e)
    return EV(0.5 * (lo + hi), 0.5 * (hi - lo))


def vmin_const(a, c):
    cc = const(c)
    return EV(np.minimum(a.v, cc.v), np.maximum(a.e, cc.e))


def absval(a):
    return EV(np.abs(a.v), a.e)


def neg(a):
    return EV(-a.v, a.e)


def select(cond_ev, when_neg, otherwise):
    """`cond < 0 ? when_neg : otherwise` by the sign BIT of the computed cond: where the error can flip the sign, the other arm's
    value may be what the device computes -- charged with the distance between the arms."""
    take = cond_ev.v < 0
    v = np.where(take, when_neg.v, otherwise.v)
    e = np.where(take, when_neg.e, otherwise.e)
    unsure = np.abs(cond_ev.v) <= cond_ev.e
    e = np.where(unsure, np.maximum(when_neg.e, otherwise.e) + np.abs(when_neg.v - otherwise.v), e)
    return EV(v, e)


def ratio7(s):
    """h2_ratio7 (MM_PRE_RATIO_ONE_RSQ): sqrt(z / (z + 1)), z = s^7, as y rsq(y (y + 1/16) + 2^-24), y = z / 16, s clamped at 3.26.
    The computed y enters numerator and denominator alike, so its error is NOT charged twice: the result is the increasing
    function f(y) = y / sqrt(y (y + 1/16) + 2^-24) of the computed y (endpoints of its interval), plus the roundings of the four
    operations after y (sum, product, sum: 3 u on the root's operand, halved by the root; the rsq's ulp; the last product)."""
    s = vmin_const(s, 3.26)
    s2 = mul(s, s)
    y = mul(mul(mul(s2, s2), s2), scale_pow2(s, 1.0 / 16.0))

    def f(t):
        t = np.maximum(t, 0.0)
        return t / np.sqrt(t * (t + 1.0 / 16.0) + 2.0 ** -24)
    fv = f(y.v)
    prop = np.maximum(np.abs(f(y.v - y.e) - fv), np.abs(f(y.v + y.e) - fv))
    # y (y + 1/16) + 2^-24 where y underflows: its absolute rounding (2 eta) against 2^-24 moves the root by a quarter at most, times y <= 2^-12
    return EV(fv, prop + (1.5 * U + ULP + U) * fv + np.minimum(4 * ETA, fv))


def lut_lookup(hx, hy, index_slack=1e-3):
    """The fp32 part of ciede2000_h2: diamond angle of (hx, hy) -> interpolated {0.0075 T, G}.  hx, hy are halfs converted exactly;
    their carried errors move the index; the fp32 arithmetic itself adds ~1e-6 of an index step."""
    fx, fy = hx.v, hy.v
    den = np.abs(fx) + np.abs(fy) + 1e-30
    qd = fx / den
    x = -(N_LUT / 4 - 2.0 ** -13) * np.copysign(1.0 + qd, fy) + N_LUT / 2
    idx = np.clip(np.floor(x).astype(np.int64), 0, N_LUT)
    fr = x - np.floor(x)
    t = LUT_T[idx] + fr * (LUT_T[idx + 1] - LUT_T[idx])
    g = LUT_G[idx] + fr * (LUT_G[idx + 1] - LUT_G[idx])
    # index error: d qd <= (|fy| e_x + |fx| e_y) / (den - e_x - e_y)^2, x = 512 (1 + qd) up to sign and offset; the circle is
    # continuous across fy = 0 (x = 0 and x = N hold the same entry), so a sign flip of fy costs no more than the distance
    room = den - hx.e - hy.e
    with np.errstate(divide="ignore", invalid="ignore"):
        dq = np.where(room > 0, (np.abs(fy) * hx.e + np.abs(fx) * hy.e) / (room * room), np.inf)
    dx = np.minimum((N_LUT / 4) * dq + index_slack, N_LUT / 2.0)      # (+ the fp32 index arithmetic; never more than half the circle)
    level = np.clip(np.ceil(np.log2(np.maximum(dx + 2.0, 1.0))).astype(np.int64), 0, 10)
    at = idx % N_LUT
    e_t = np.minimum(dx * SLOPES_T[level, at], LUT_T.max() - LUT_T.min()) + 1e-9
    e_g = np.minimum(dx * SLOPES_G[level, at], LUT_G.max() - LUT_G.min()) + 1e-7
    return EV(t, e_t), EV(g, e_g)


def prepared(lab):
    """interleave_kernel + interleave_h_kernel: L / 2 - 25, a / 4, b / 4, C / 4 as halfs.  The exact counterparts are the
    unrounded quantities; the carried error is the rounding to half (and the fp32 roundings before it, 3 ulp of fp32)."""
    L, a, b = lab[..., 0], lab[..., 1], lab[..., 2]
    c = np.hypot(a, b)

    def to_half(x):
        return EV(x, U * np.abs(x) + ETA + 4e-7 * np.abs(x))
    return to_half(0.5 * L - 25.0), to_half(0.25 * a), to_half(0.25 * b), to_half(0.25 * c)


def pick(obtuse, when_obtuse, otherwise):
    return EV(np.where(obtuse, when_obtuse.v, otherwise.v), np.where(obtuse, when_obtuse.e, otherwise.e))


def form_given_g(one_g, pl, pa, pb, pc, ql, qa, qb, qc, force_obtuse=None):
    """ciede2000_h2 from `one_g` on, with one_g taken as the half the device holds (no carried error: pretier_bound runs this at
    both ends of one_g's interval -- one_g multiplies a1 AND a2, and charged independently to both its error would not cancel in
    their cross product the way it does on the device).  force_obtuse: the arm every select on dot's sign takes (the device
    takes ONE arm for all of them; where dot's error can flip the sign pretier_bound runs both and compares the results, which
    agree at dot = 0: the form is continuous there)."""
    one, half_c = EV(1.0), EV(0.5)
    a1, a2 = mul(one_g, pa), mul(one_g, qa)
    c1p = vmax(monotone(add(mul(a1, a1), mul(pb, pb)), np.sqrt), pc)
    c2p = vmax(monotone(add(mul(a2, a2), mul(qb, qb)), np.sqrt), qc)
    dot = add(mul(a1, a2), mul(pb, qb))
    cross = sub(mul(a1, qb), mul(pb, a2))
    e = add(add(mul(c1p, c2p), EV(2.0 ** -13)), absval(dot))
    # seam: obtuse and |cross| < e / 128 -> the pair counts 0 (an under-estimate: always allowed); and so does a pair with a
    # stored chroma below TINY_CHROMA on either side (squares of a / 4, b / 4 below 2^-14 are subnormal halfs: C' and with it the
    # hue terms lose their relative accuracy there, and a grey's hue terms are small but not zero beside a saturated colour)
    seam = ((dot.v < 0) & (np.abs(cross.v) < e.v / 128.0)) | (pc.v < TINY_CHROMA) | (qc.v < TINY_CHROMA)
    # (the computed e is a sum of non-negative halfs and 2^-13: never below 2^-13 less its own roundings)
    e_floor = 2.0 ** -13 * (1.0 - 4 * U)
    r = monotone(scale_pow2(e, 0.5), lambda x: 1.0 / np.sqrt(x), floor=0.5 * e_floor)
    sign_cross = np.where(cross.v < 0, -1.0, 1.0)
    obtuse = (dot.v < 0) if force_obtuse is None else np.full(dot.v.shape, bool(force_obtuse))
    # acute arm: cross r.  Obtuse arm: copysign(e, cross) r = sign sqrt(2 e) -- the computed e on both sides of the product, an
    # increasing function of it (interval endpoints; the rsq's ulp and the product's rounding on top).  Off the seam
    # |cross| >= e / 128: unless cross's error reaches that far its sign is certain; where it does, both signs are charged
    d_acute = mul(cross, r)
    root = monotone(scale_pow2(e, 2.0), np.sqrt, rel=ULP + U, floor=2.0 * e_floor)
    d_obtuse = EV(sign_cross * root.v, np.where(np.abs(cross.v) <= cross.e, 2.0 * root.mag(), root.e))
    d_h = pick(obtuse, d_obtuse, d_acute)
    c2s = pick(obtuse, neg(c2p), c2p)
    wx = add(mul(a2, c1p), mul(a1, c2s))
    wy = add(mul(qb, c1p), mul(pb, c2s))
    # acute: (wx, wy); obtuse: sign(cross) (wy, -wx)
    ox = EV(sign_cross * wy.v, wy.e)
    oy = EV(-sign_cross * wx.v, wx.e)
    hx, hy = pick(obtuse, ox, wx), pick(obtuse, oy, wy)
    tt, ss = lut_lookup(hx, hy)
    t4 = rounded(4.0 * tt.v, 4.0 * tt.e)            # half(4 tt)
    s2 = rounded(ss.v, ss.e)
    csum = add(c1p, c2p)
    r_t = mul(s2, ratio7(mul(csum, const(0.08))))
    lb = add(pl, ql)
    u_l = add(mul(lb, lb), EV(20.0))
    s_l = add(mul(sub(mul(const(0.0075), u_l), const(0.15)), monotone(u_l, lambda x: 1.0 / np.sqrt(x), floor=1.0)), half_c)
    s_c = add(mul(const(0.09), csum), one)
    s_h = add(mul(csum, t4), one)
    big_d = mul(s_c, s_h)
    # inv_all = rcp(big_d s_l); inv_d = (inv_all s_l) 4; t_l's factor = inv_all big_d; t_c = ((c2p - c1p) s_h) inv_d;
    # t_h = (d_h s_c) inv_d.  A computed factor that appears in a reciprocal's operand AND outside it cancels on the device as in
    # exact arithmetic -- s_l and big_d in t_l's factor, s_h in t_c, s_c in t_h -- so its own error is not charged, only the
    # roundings of the operations it passes through: t_l = dl / s_l, t_c = 4 dc / s_c, t_h = 4 d_h / s_h, each with the product's,
    # the reciprocal's and the following products' roundings (3 u + one ulp for t_l, 5 u + one ulp for the other two)
    d_l, d_c = sub(ql, pl), sub(c2p, c1p)

    def quotient(num, den, scale, rel, floor):
        # (the computed denominator is 1 + a product of non-negative halfs, or 0.5 + one: never below `floor`)
        lo = np.maximum(den.v - den.e, floor)
        val = scale * num.v / den.v
        err = scale * (num.e / lo + np.abs(num.v) * den.e / (lo * den.v))
        return EV(val, err + rel * (np.abs(val) + err) + ETA)
    t_l = quotient(d_l, s_l, 1.0, 3 * U + ULP, 0.5 * (1 - 2 * U))
    t_c = quotient(d_c, s_c, 4.0, 5 * U + ULP, 1 - 2 * U)
    t_h = quotient(d_h, s_h, 4.0, 5 * U + ULP, 1 - 2 * U)
    # xx = t_l^2 + (t_c (r_t t_h + t_c) + t_h^2): the partial derivatives of the quadratic form, not the sequence of operations
    # -- near the blue hues r_t -> -1.7 and t_c^2 + t_h^2 + r_t t_c t_h cancels to a fraction of its terms, which the
    # sequence's own intermediate errors would be charged in full for although they cancel the same way --, second-order terms in
    # full, and the five roundings at the size of the terms they round
    first = (np.abs(2 * t_l.v) * t_l.e + np.abs(2 * t_c.v + r_t.v * t_h.v) * t_c.e + np.abs(2 * t_h.v + r_t.v * t_c.v) * t_h.e +
             np.abs(t_c.v * t_h.v) * r_t.e)
    second = (t_l.e ** 2 + t_c.e ** 2 + t_h.e ** 2 + np.abs(r_t.v) * t_c.e * t_h.e +
              r_t.e * (np.abs(t_c.v) * t_h.e + np.abs(t_h.v) * t_c.e + t_c.e * t_h.e))
    terms = t_l.v ** 2 + t_c.v ** 2 + t_h.v ** 2 + np.abs(r_t.v * t_c.v * t_h.v)
    xx_v = np.maximum(t_l.v ** 2 + t_c.v ** 2 + t_h.v ** 2 + r_t.v * t_c.v * t_h.v, 0.0)
    xx = EV(xx_v, first + second + 4 * U * (terms + first + second) + 4 * ETA)
    v = monotone(xx, np.sqrt)
    overflow = np.zeros(v.v.shape, dtype=bool)
    for t in (a1, a2, c1p, c2p, dot, cross, e, d_h, wx, wy, csum, u_l, big_d, t_l, t_c, t_h, xx):
        overflow |= ~(t.mag() < HMAX)
    return v.v, np.where(overflow, np.inf, v.e), seam, np.abs(dot.v) <= dot.e


def pretier_bound(p, q):
    """The pre-tier's form on the pair (p, q) [..., 3] f32-valued Lab: (value in exact arithmetic, carried error bound, seam mask).
    Follows ciede2000_h2 line by line (default build: chroma floor, one rsq per ratio, one reciprocal)."""
    pl, pa, pb, pc = prepared(p)
    ql, qa, qb, qc = prepared(q)
    one_g = sub(EV(1.5), scale_pow2(ratio7(mul(add(pc, qc), const(0.08))), 0.5))
    return bound_of(form_given_g, one_g, (pl, pa, pb, pc, ql, qa, qb, qc))


# --------------------------------------------------------------------------- the fp32 filter form (ciede2000_fast<PREP = true>)
def ratio7_single(s):
    """sqrt_ratio7_scaled: z = s^7 + 1e-30, z rsq(z z + z) -- as ratio7 above: an increasing function of the computed z."""
    s2 = mul(s, s)
    z = add(mul(mul(mul(s2, s2), s2), s), EV(1e-30))

    def f(t):
        t = np.maximum(t, 1e-30)
        return t / np.sqrt(t * t + t)
    fv = f(z.v)
    prop = np.maximum(np.abs(f(z.v - z.e) - fv), np.abs(f(z.v + z.e) - fv))
    return EV(fv, prop + (1.5 * U + ULP + U) * fv)


def prepared_single(lab):
    """interleave_kernel: L / 2 - 25 (one fma), a, b as handed over, C / 50 (fma, sqrt, one product)."""
    L, a, b = lab[..., 0], lab[..., 1], lab[..., 2]
    c = np.hypot(a, b) * 0.02
    return EV(0.5 * L - 25.0, U * np.abs(0.5 * L - 25.0)), EV(a), EV(b), EV(c, (2 * U + ULP + 2 * U) * c)


def filter_form_given_g(one_g, pl, pa, pb, pc, ql, qa, qb, qc, force_obtuse=None):
    """ciede2000_fast<true> from one_g on (colour_diff_fast.hpp:97-172), the way form_given_g follows ciede2000_h2."""
    one, half_c = EV(1.0), EV(0.5)
    a1, a2 = mul(one_g, pa), mul(one_g, qa)
    c1p = monotone(add(add(mul(a1, a1), mul(pb, pb)), EV(1e-36)), np.sqrt)
    c2p = monotone(add(add(mul(a2, a2), mul(qb, qb)), EV(1e-36)), np.sqrt)
    dot = add(mul(a1, a2), mul(pb, qb))
    cross = sub(mul(a1, qb), mul(pb, a2))
    e = add(mul(c1p, c2p), absval(dot))
    # needs_exact: obtuse and |cross| < 1.50001e-5 e -> the pair is scored by the fp64 form (never judged here)
    handed_over = (dot.v < 0) & (np.abs(cross.v) < 1.50001e-5 * e.v)
    r = monotone(scale_pow2(e, 0.5), lambda x: 1.0 / np.sqrt(x), floor=0.25e-36)
    sign_cross = np.where(cross.v < 0, -1.0, 1.0)
    obtuse = (dot.v < 0) if force_obtuse is None else np.full(dot.v.shape, bool(force_obtuse))
    d_acute = mul(cross, r)
    root = monotone(scale_pow2(e, 2.0), np.sqrt, rel=ULP + U, floor=1e-36)
    d_obtuse = EV(sign_cross * root.v, np.where(np.abs(cross.v) <= cross.e, 2.0 * root.mag(), root.e))
    d_h = pick(obtuse, d_obtuse, d_acute)
    c2s = pick(obtuse, neg(c2p), c2p)
    wx = add(add(mul(a2, c1p), mul(a1, c2s)), EV(1e-30))
    wy = add(mul(qb, c1p), mul(pb, c2s))
    ox = EV(sign_cross * wy.v, wy.e)
    oy = EV(-sign_cross * wx.v, wx.e)
    hx, hy = pick(obtuse, ox, wx), pick(obtuse, oy, wy)
    tt, ss = lut_lookup(hx, hy, index_slack=N_LUT / 4 * 8 * U + 2.0 ** -13)      # (the index arithmetic is fp32 here: a few ulp of 1 + qd)
    t, s2 = rounded(tt.v, tt.e), rounded(ss.v, ss.e)
    csum = add(c1p, c2p)
    r_t = mul(s2, ratio7_single(mul(csum, const(0.02))))
    lb = add(pl, ql)
    u_l = add(mul(lb, lb), EV(20.0))
    s_l = add(mul(sub(mul(const(0.0075), u_l), const(0.15)), monotone(u_l, lambda x: 1.0 / np.sqrt(x), floor=1.0)), half_c)
    s_c = add(mul(const(0.0225), csum), one)
    s_h = add(mul(csum, t), one)
    d_l, d_c = sub(ql, pl), sub(c2p, c1p)

    def quotient(num, den, rel, floor):
        lo = np.maximum(den.v - den.e, floor)
        val = num.v / den.v
        err = num.e / lo + np.abs(num.v) * den.e / (lo * den.v)
        return EV(val, err + rel * (np.abs(val) + err))
    # t_l = d_l rcp(s_l); tC = (d_c s_h) / (s_c s_h), tH = (d_h s_c) / (s_c s_h): the computed s_h (s_c) cancels as in ciede2000_h2
    t_l = quotient(d_l, s_l, U + ULP, 0.5 * (1 - 2 * U))
    t_c = quotient(d_c, s_c, 3 * U + ULP, 1 - 2 * U)
    t_h = quotient(d_h, s_h, 3 * U + ULP, 1 - 2 * U)
    first = (np.abs(2 * t_l.v) * t_l.e + np.abs(2 * t_c.v + r_t.v * t_h.v) * t_c.e + np.abs(2 * t_h.v + r_t.v * t_c.v) * t_h.e +
             np.abs(t_c.v * t_h.v) * r_t.e)
    second = (t_l.e ** 2 + t_c.e ** 2 + t_h.e ** 2 + np.abs(r_t.v) * t_c.e * t_h.e +
              r_t.e * (np.abs(t_c.v) * t_h.e + np.abs(t_h.v) * t_c.e + t_c.e * t_h.e))
    terms = t_l.v ** 2 + t_c.v ** 2 + t_h.v ** 2 + np.abs(r_t.v * t_c.v * t_h.v)
    xx_v = np.maximum(t_l.v ** 2 + t_c.v ** 2 + t_h.v ** 2 + r_t.v * t_c.v * t_h.v, 0.0)
    xx = EV(xx_v, first + second + 6 * U * (terms + first + second))
    v = monotone(xx, np.sqrt)
    return v.v, v.e, handed_over, np.abs(dot.v) <= dot.e


# --------------------------------------------------------------------------- the algebraic fp64 form (colour_diff.hpp ciede2000_algebraic)
def ciede2000_ref_ld(p, q):
    """CIEDE2000 in long double: the yardstick of the fp64 form (float64 itself is what is being measured there)."""
    ld = np.longdouble
    p, q = p.astype(ld), q.astype(ld)
    pi = ld("3.14159265358979323846264338327950288")
    L1, a1, b1 = p[..., 0], p[..., 1], p[..., 2]
    L2, a2, b2 = q[..., 0], q[..., 1], q[..., 2]
    k25 = ld(25) ** 7
    c1, c2 = np.sqrt(a1 * a1 + b1 * b1), np.sqrt(a2 * a2 + b2 * b2)
    cb = (c1 + c2) / 2
    cb7 = cb ** 7
    g = (1 - np.sqrt(cb7 / (cb7 + k25))) / 2
    a1p, a2p = (1 + g) * a1, (1 + g) * a2
    c1p, c2p = np.sqrt(a1p * a1p + b1 * b1), np.sqrt(a2p * a2p + b2 * b2)

    def hue(b, ap):
        h = np.arctan2(b, ap)
        h = np.where(h < 0, h + 2 * pi, h)
        return np.where((b == 0) & (ap == 0), ld(0), h)
    h1, h2 = hue(b1, a1p), hue(b2, a2p)
    dl, dc = L2 - L1, c2p - c1p
    dh = h2 - h1
    zero = (c1p * c2p) == 0
    dh = np.where(zero, ld(0), np.where(dh > pi, dh - 2 * pi, np.where(dh < -pi, dh + 2 * pi, dh)))
    dH = 2 * np.sqrt(c1p * c2p) * np.sin(dh / 2)
    lb, cbp = (L1 + L2) / 2, (c1p + c2p) / 2
    hs = h1 + h2
    hb = np.where(zero, hs, np.where(np.abs(h1 - h2) <= pi, hs / 2, np.where(hs < 2 * pi, (hs + 2 * pi) / 2, (hs - 2 * pi) / 2)))
    rad = lambda d: ld(d) * pi / 180
    T = 1 - ld("0.17") * np.cos(hb - rad(30)) + ld("0.24") * np.cos(2 * hb) + ld("0.32") * np.cos(3 * hb + rad(6)) - ld("0.20") * np.cos(4 * hb - rad(63))
    dth = rad(30) * np.exp(-(((hb - rad(275)) / rad(25)) ** 2))
    cbp7 = cbp ** 7
    rc = 2 * np.sqrt(cbp7 / (cbp7 + k25))
    sl = 1 + ld("0.015") * (lb - 50) ** 2 / np.sqrt(20 + (lb - 50) ** 2)
    sc = 1 + ld("0.045") * cbp
    sh = 1 + ld("0.015") * cbp * T
    rt = -np.sin(2 * dth) * rc
    tl, tc, th = dl / sl, dc / sc, dH / sh
    return np.sqrt(np.maximum(tl * tl + tc * tc + th * th + rt * tc * th, 0))


def ratio7_double(c_bar):
    """x rsqrt(x (x + 25^7) + 1e-300), x = c_bar^7: an increasing function of the computed x."""
    c2 = mul(c_bar, c_bar)
    x = mul(mul(c2, c2), mul(c2, c_bar))
    k = DT(25) ** 7

    def f(t):
        t = np.maximum(t, 0)
        return t / np.sqrt(t * (t + k) + DT(1e-300))
    fv = f(x.v)
    prop = np.maximum(np.abs(f(x.v - x.e) - fv), np.abs(f(x.v + x.e) - fv))
    return EV(fv, prop + (1.5 * U + ULP + U) * fv)


def algebraic_form_given_g(one_g, L1, a1, b1, L2, a2, b2, force_obtuse=None):
    """ciede2000_algebraic from one_g on.  Every library call is charged two ulp (ocml's documented bounds: sqrt and division are
    correctly rounded, rsqrt / exp one ulp, atan2 two)."""
    a1p, a2p = mul(one_g, a1), mul(one_g, a2)
    c1p = monotone(add(mul(a1p, a1p), mul(b1, b1)), np.sqrt)
    c2p = monotone(add(mul(a2p, a2p), mul(b2, b2)), np.sqrt)
    prod = mul(c1p, c2p)
    dot = add(mul(a1p, a2p), mul(b1, b2))
    cross = sub(mul(a1p, b2), mul(b1, a2p))
    grey = ~(prod.v > 0)
    e = add(prod, absval(dot))
    handed_over = ~grey & (dot.v < 0) & (np.abs(cross.v) <= 1e-9 * e.v)
    safe_e = EV(np.where(grey, DT(1), e.v), np.where(grey, DT(0), e.e))
    r = monotone(scale_pow2(safe_e, 0.5), lambda x: 1 / np.sqrt(x), floor=DT(1e-300))
    sign_cross = np.where(cross.v < 0, DT(-1), DT(1))
    obtuse = (dot.v < 0) if force_obtuse is None else np.full(dot.v.shape, bool(force_obtuse))
    d_acute = mul(cross, r)
    root = monotone(scale_pow2(safe_e, 2.0), np.sqrt, rel=ULP + U)
    d_obtuse = EV(sign_cross * root.v, np.where(np.abs(cross.v) <= cross.e, 2 * root.mag(), root.e))
    d_h = pick(obtuse, d_obtuse, d_acute)
    d_h = EV(np.where(grey, DT(0), d_h.v), np.where(grey, DT(0), d_h.e))
    c2s = pick(obtuse, neg(c2p), c2p)
    wx = add(mul(a2p, c1p), mul(a1p, c2s))
    wy = add(mul(b2, c1p), mul(b1, c2s))
    hx = pick(obtuse, EV(sign_cross * wy.v, wy.e), wx)
    hy = pick(obtuse, EV(-sign_cross * wx.v, wx.e), wy)
    first_grey = ~(c1p.v > 0)
    hx = EV(np.where(grey, np.where(first_grey, a2p.v, a1p.v), hx.v), np.where(grey, np.where(first_grey, a2p.e, a1p.e), hx.e))
    hy = EV(np.where(grey, np.where(first_grey, b2.v, b1.v), hy.v), np.where(grey, np.where(first_grey, b2.e, b1.e), hy.e))
    both = ~(c1p.v > 0) & ~(c2p.v > 0)
    hx = EV(np.where(both, DT(1), hx.v), np.where(both, DT(0), hx.e))
    hy = EV(np.where(both, DT(0), hy.v), np.where(both, DT(0), hy.e))
    # the mean-hue direction as an angle and its error (the normalisation cancels a common factor of hx, hy exactly)
    n2 = hx.v * hx.v + hy.v * hy.v
    room = np.sqrt(n2) - hx.e - hy.e
    with np.errstate(divide="ignore", invalid="ignore"):
        d_ang = np.where(room > 0, (np.abs(hy.v) * hx.e + np.abs(hx.v) * hy.e) / (room * room), np.inf) + 4 * U
    theta = np.arctan2(hy.v, hx.v)
    ch, sh = np.cos(theta), np.sin(theta)
    handed_over = handed_over | (~grey & (ch > 0) & (np.abs(sh) <= 1e-9))
    rad = lambda d: DT(d) * DT(np.pi) / 180
    t_exact = (1 - DT("0.17") * np.cos(theta - rad(30)) + DT("0.24") * np.cos(2 * theta) + DT("0.32") * np.cos(3 * theta + rad(6))
               - DT("0.20") * np.cos(4 * theta - rad(63)))
    t = EV(t_exact, np.minimum(DT("2.41") * d_ang, DT(3)) + 24 * U)       # the multiple-angle identities: ~20 operations on values <= 1
    h_bar = np.where(theta < 0, theta + 2 * DT(np.pi), theta)
    z = (h_bar - rad(275)) / rad(25)
    ez = np.exp(-(z * z))
    e_z = np.minimum(2 * np.abs(z) * ez / rad(25) * (d_ang + 4 * U * 7), DT(1)) + 4 * U * (1 + z * z) * ez     # atan2 two ulp of <= 2 pi; exp one ulp, its operand's rounding
    x2 = 2 * rad(30) * ez
    sn = EV(np.sin(x2), 2 * rad(30) * e_z + 12 * U)          # the Taylor polynomial: 2e-17 truncation, nine fmas
    cp_bar = scale_pow2(add(c1p, c2p), 0.5)
    r_t = neg(mul(sn, scale_pow2(ratio7_double(cp_bar), 2.0)))
    l_off = sub(scale_pow2(add(L1, L2), 0.5), EV(50.0))
    l2 = mul(l_off, l_off)
    s_l = add(EV(1.0), mul(mul(const(0.015), l2), monotone(add(EV(20.0), l2), lambda x: 1 / np.sqrt(x), floor=DT(1))))
    s_c = add(mul(const(0.045), cp_bar), EV(1.0))
    s_h = add(mul(mul(const(0.015), cp_bar), t), EV(1.0))
    d_l, d_c = sub(L2, L1), sub(c2p, c1p)

    def quotient(num, den, rel, floor):
        lo = np.maximum(den.v - den.e, floor)
        val = num.v / den.v
        err = num.e / lo + np.abs(num.v) * den.e / (lo * den.v)
        return EV(val, err + rel * (np.abs(val) + err))
    # one division for the three terms: the factors that appear on both sides cancel on the device as in exact arithmetic
    t_l = quotient(d_l, s_l, 6 * U, 1 - 2 * U)
    t_c = quotient(d_c, s_c, 6 * U, 1 - 2 * U)
    t_h = quotient(d_h, s_h, 6 * U, 1 - 2 * U)
    first = (np.abs(2 * t_l.v) * t_l.e + np.abs(2 * t_c.v + r_t.v * t_h.v) * t_c.e + np.abs(2 * t_h.v + r_t.v * t_c.v) * t_h.e +
             np.abs(t_c.v * t_h.v) * r_t.e)
    second = (t_l.e ** 2 + t_c.e ** 2 + t_h.e ** 2 + np.abs(r_t.v) * t_c.e * t_h.e +
              r_t.e * (np.abs(t_c.v) * t_h.e + np.abs(t_h.v) * t_c.e + t_c.e * t_h.e))
    terms = t_l.v ** 2 + t_c.v ** 2 + t_h.v ** 2 + np.abs(r_t.v * t_c.v * t_h.v)
    xx_v = np.maximum(t_l.v ** 2 + t_c.v ** 2 + t_h.v ** 2 + r_t.v * t_c.v * t_h.v, 0)
    xx = EV(xx_v, first + second + 5 * U * (terms + first + second))
    v = monotone(xx, np.sqrt, rel=U)
    return v.v, v.e, handed_over, ~grey & (np.abs(dot.v) <= dot.e)


def algebraic_bound(p, q):
    """The algebraic fp64 form on the pair: (value in long double, carried error bound, pairs handed to the reference form)."""
    set_precision("double")
    try:
        L1, a1, b1 = (EV(p[..., i]) for i in range(3))
        L2, a2, b2 = (EV(q[..., i]) for i in range(3))
        c1 = monotone(add(mul(a1, a1), mul(b1, b1)), np.sqrt, rel=U)
        c2 = monotone(add(mul(a2, a2), mul(b2, b2)), np.sqrt, rel=U)
        one_g = sub(EV(1.5), scale_pow2(ratio7_double(scale_pow2(add(c1, c2), 0.5)), 0.5))
        v, e, handed, = bound_of(algebraic_form_given_g, one_g, (L1, a1, b1, L2, a2, b2))
        return v.astype(np.float64), e.astype(np.float64), handed
    finally:
        set_precision("half")


def bound_of(form, one_g, args):
    """The form at the middle and both ends of one_g's interval, and on either arm where dot's sign is within its error."""
    v0, e0, seam, _ = form(EV(one_g.v), *args)
    worst = np.zeros_like(v0)
    for g in (one_g.v, one_g.v - one_g.e, one_g.v + one_g.e):
        v1, e1, seam1, unsure = form(EV(g), *args)
        worst = np.maximum(worst, np.abs(v1 - v0) + e1)
        seam = seam & seam1          # (counted as a seam pair only where every one_g of the interval says so)
        if unsure.any():
            for arm in (False, True):
                v2, e2, _, _ = form(EV(g), *args, force_obtuse=arm)
                worst = np.where(unsure, np.maximum(worst, np.abs(v2 - v0) + e2), worst)
    return v0, worst, seam


def filter_bound(p, q):
    """The fp32 filter form on the pair: (value in exact arithmetic, carried error bound, pairs handed to the fp64 form)."""
    set_precision("single")
    try:
        pl, pa, pb, pc = prepared_single(p)
        ql, qa, qb, qc = prepared_single(q)
        one_g = sub(EV(1.5), scale_pow2(ratio7_single(add(pc, qc)), 0.5))
        return bound_of(filter_form_given_g, one_g, (pl, pa, pb, pc, ql, qa, qb, qc))
    finally:
        set_precision("half")


def euclid_bound(p, q):
    """euclidean_fast (RGB / CIE76 filter form): sqrt(fma(d0, d0, fma(d1, d1, d2 d2))), d_i = p_i - q_i in fp32.
    Returns (value in exact arithmetic, carried error bound); the engine trusts it with 5e-4 absolute per pixel pair."""
    set_precision("single")
    try:
        d = [sub(EV(p[..., i]), EV(q[..., i])) for i in range(3)]
        m = add(mul(d[0], d[0]), add(mul(d[1], d[1]), mul(d[2], d[2])))
        v = monotone(m, np.sqrt)
        return v.v, v.e
    finally:
        set_precision("half")


# --------------------------------------------------------------------------- input families
def lab_of_u8(bgr):
    """synth.py's sRGB -> Lab (D65), the conversion the engine's u8 setters specify; float32-valued."""
    sys.path.insert(0, ROOT)
    from mosaicmagnifique_amd import synth
    return synth.bgr_u8_to_lab_f32(np.ascontiguousarray(bgr.reshape(-1, 1, 3))).reshape(-1, 3).astype(np.float64)


def polar(L, c, h):
    return np.stack([L, c * np.cos(h), c * np.sin(h)], axis=-1).astype(np.float32).astype(np.float64)


def families(n, rng):
    """name -> (p, q): the zones where the form is weakest and the inputs real jobs hold."""
    out = {}
    two_pi = 2 * math.pi
    h1, h2 = rng.uniform(0, two_pi, n), rng.uniform(0, two_pi, n)
    chroma_levels = np.array([0.0, 1e-6, 1e-5, 1e-4, 1e-3, 1e-2, 0.05, 0.2, 0.5, 1, 2, 4, 8, 15, 25, 40, 60, 80, 100, 120, 135, 150])
    c1, c2 = rng.choice(chroma_levels, n), rng.choice(chroma_levels, n)
    L1, L2 = rng.uniform(0, 100, n), rng.uniform(0, 100, n)
    out["chroma levels x any hue"] = (polar(L1, c1, h1), polar(L2, c2, h2))
    c = rng.uniform(0, 150, n)
    out["uniform in the box"] = (polar(L1, c, h1), polar(L2, rng.uniform(0, 150, n), h2))
    # near-identical colours (what a contender's pixels are)
    d = rng.normal(0, 1.0, (n, 3)) * rng.choice([1e-3, 1e-2, 0.1, 1.0, 3.0], (n, 1))
    base = polar(L1, c, h1)
    out["near-identical"] = (base, (base + d).astype(np.float32).astype(np.float64))
    # similar colours at the distance of a good candidate (dE 5 .. 25)
    d = rng.normal(0, 1.0, (n, 3)) * rng.uniform(3, 15, (n, 1))
    near = base + d
    near[:, 0] = np.clip(near[:, 0], 0, 100)
    out["contender-like"] = (base, near.astype(np.float32).astype(np.float64))
    # greys as a float conversion leaves them, beside saturated colours (the maximiser's find of round 4)
    tiny = 10.0 ** rng.uniform(-7, -1, n)
    out["tiny chroma x saturated"] = (polar(L1, tiny, h1), polar(L2, rng.uniform(60, 150, n), h2))
    out["tiny chroma x tiny chroma"] = (polar(L1, tiny, h1), polar(L2, 10.0 ** rng.uniform(-7, -1, n), h2))
    # both sides of the opposite-hue seam, and the quarter turn where the arms of the form change
    off = rng.choice([-1, 1], n) * 10.0 ** rng.uniform(-6, -0.5, n)
    out["around the seam"] = (polar(L1, c, h1), polar(L2, rng.uniform(0.5, 150, n), h1 + math.pi + off))
    out["around the quarter turn"] = (polar(L1, c, h1), polar(L2, rng.uniform(0.5, 150, n), h1 + math.pi / 2 + off))
    # saturated blue, where R_T peaks
    hb = math.radians(275) + rng.normal(0, math.radians(12), n)
    out["saturated blue"] = (polar(L1, rng.uniform(40, 150, n), hb), polar(L2, rng.uniform(40, 150, n), hb + rng.normal(0, 0.3, n)))
    # the corners of the box
    out["corners"] = (polar(rng.choice([0.0, 100.0], n), rng.choice([0.0, 150.0], n), h1),
                      polar(rng.choice([0.0, 100.0], n), rng.choice([0.0, 150.0], n), h2))
    # what real jobs hold: colours of u8 images
    u1, u2 = rng.integers(0, 256, (n, 3), dtype=np.uint8), rng.integers(0, 256, (n, 3), dtype=np.uint8)
    out["u8 colours, random pairs"] = (lab_of_u8(u1), lab_of_u8(u2))
    du = np.clip(u1.astype(np.int64) + rng.integers(-24, 25, (n, 3)), 0, 255).astype(np.uint8)
    out["u8 colours, similar pairs"] = (lab_of_u8(u1), lab_of_u8(du))
    grey = np.repeat(rng.integers(0, 256, (n, 1), dtype=np.uint8), 3, axis=1)
    out["u8 greys x u8 colours"] = (lab_of_u8(grey), lab_of_u8(u2))
    return out


def evaluate(p, q, form="pretier"):
    """Per pair: true value, centre - true, the carried bound, the bound on what the band must hold, pairs not judged.
    pretier: the OVER-estimate (v - true) + eps (an under-estimate only delays an exit; a seam pair counts 0);
    filter: |v - true| + eps (the contender selection trusts the band on both sides; seam pairs go to the fp64 form)."""
    if form == "algebraic":
        true = ciede2000_ref_ld(p, q)
        v, e, seam = algebraic_bound(p, q)
        need = np.where(seam, 0.0, np.abs(v - true.astype(np.float64)) + e)        # (centre and yardstick are long doubles: their difference is the method error)
        return true.astype(np.float64), v - true.astype(np.float64), e, need, seam
    true = ciede2000_ref(p, q)
    if form == "pretier":
        v, e, seam = pretier_bound(p, q)
        need = np.where(seam, -true, (v - true) + e)
    else:
        v, e, seam = filter_bound(p, q)
        need = np.where(seam, 0.0, np.abs(v - true) + e)
    return true, v - true, e, need, seam


def frontier(true, need, abs_parts):
    """The smallest relative part that holds every pair for a few absolute parts."""
    rows = []
    for a in abs_parts:
        rel = np.maximum(need - a, 0.0) / np.maximum(true, 1e-9)
        rows.append((a, float(rel.max())))
    return rows


# what the engine ships: filter.hip kPreAbs / kPreRel, engine.hip filter_band (per pixel pair)
BANDS = {"pretier": (0.45, 3.8e-2), "filter": (1e-3, 6e-5), "algebraic": (2e-12, 0.0)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--form", choices=("pretier", "filter", "algebraic"), default="pretier")
    ap.add_argument("--pairs", type=int, default=400000)
    ap.add_argument("--abs", type=float, default=None)
    ap.add_argument("--rel", type=float, default=None)
    ap.add_argument("--seed", type=int, default=5)
    args = ap.parse_args()
    band_abs = BANDS[args.form][0] if args.abs is None else args.abs
    band_rel = BANDS[args.form][1] if args.rel is None else args.rel
    rng = np.random.default_rng(args.seed)
    all_true, all_need = [], []
    print("%s form, band %.3g + %.3g v; %d pairs per family" % (args.form, band_abs, band_rel, args.pairs))
    print("%-28s %9s %9s %11s %11s %9s" % ("family", "max share", "at dE", "centre-true", "carried", "unjudged %"))
    for name, (p, q) in families(args.pairs, rng).items():
        true, meth, e, need, seam = evaluate(p, q, args.form)
        share = need / (band_abs + band_rel * true)
        k = int(np.argmax(share))
        print("%-28s %9.3f %9.2f %11.3g %11.3g %9.3f" % (name, share[k], true[k], meth[k], e[k], 100.0 * seam.mean()))
        all_true.append(true)
        all_need.append(need)
    true, need = np.concatenate(all_true), np.concatenate(all_need)
    print("frontier (abs -> smallest rel that holds every pair):")
    parts = {"pretier": (0.25, 0.3, 0.35, 0.4, 0.45, 0.5, 0.6, 0.8, 1.0), "filter": (5e-5, 1e-4, 2e-4, 3e-4, 5e-4, 1e-3),
             "algebraic": (1e-14, 1e-13, 5e-13, 1e-12)}[args.form]
    for a, r in frontier(true, need, parts):
        print("   %.3g + %.3g v" % (a, r))


if __name__ == "__main__":
    main()
