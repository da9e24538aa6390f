// colour_diff.hpp -- per-pixel colour differences on gfx950, fp64 "reference form".
//
// These are the arbiter versions: same operation order as the reference's CPU
// functions (src/Photomosaic/ColourDifference.cpp:28-33 and :42-158), all fp64,
// and the translation unit is built with -ffp-contract=off so no a*b+c is fused.
// The reference's own CUDA device functions (CUDA/ColourDifference.cuh:24-139)
// do a few products in fp32 first and therefore differ from its CPU path in the
// last bits; the parity contract here is the CPU path.
//
// For RGB/CIE76 the result is bit-identical to the CPU: differences of
// f32-valued doubles and their squares are computed with the same IEEE
// operations and the fp64 sqrt lowering on gfx950 is correctly rounded.
// For CIEDE2000 the only deviations are the ocml vs host-libm roundings of
// atan2 / sin / cos / exp (<= 1-2 ulp each).
#pragma once

#include <hip/hip_runtime.h>

namespace mm {

enum : int { RGB_EUCLIDEAN = 0, CIE76 = 1, CIEDE2000 = 2 };

struct Px { double c0, c1, c2; };   // BGR or Lab triplet promoted f32 -> f64 (Vec3f -> Vec3d)

// ColourDifference::calculateRGBEuclidean, ColourDifference.cpp:28-33.
__device__ __forceinline__ double euclidean(const Px& p, const Px& q)
{
    const double d0 = p.c0 - q.c0, d1 = p.c1 - q.c1, d2 = p.c2 - q.c2;
    return sqrt(d0 * d0 + d1 * d1 + d2 * d2);
}

__device__ __forceinline__ double seventh_power(double x) { return x * x * x * x * x * x * x; }

// Hue angle in [0, 2pi), ColourDifference.cpp:61-83.
__device__ __forceinline__ double hue_angle(double b, double a_prime, double two_pi)
{
    double h = atan2(b, a_prime);
    if (h < 0.0) h += two_pi;
    return (b == 0.0 && a_prime == 0.0) ? 0.0 : h;
}

// ColourDifference::calculateCIEDE2000, ColourDifference.cpp:42-158.
// p = main-image (cell) pixel, q = library pixel (call order of
// CPUPhotomosaicGenerator.cpp:159).
__device__ __forceinline__ double ciede2000(const Px& p, const Px& q)
{
    constexpr double kPi = 3.14159265358979323846;
    constexpr double two_pi = (360.0 * kPi) / 180.0, pi = (180.0 * kPi) / 180.0;
    constexpr double rad6 = (6.0 * kPi) / 180.0, rad25 = (25.0 * kPi) / 180.0;
    constexpr double rad30 = (30.0 * kPi) / 180.0, rad63 = (63.0 * kPi) / 180.0;
    constexpr double rad275 = (275.0 * kPi) / 180.0, rad550 = (550.0 * kPi) / 180.0;
    constexpr double twentyfive_7 = 6103515625.0;

    const double c1 = sqrt((p.c1 * p.c1) + (p.c2 * p.c2));
    const double c2 = sqrt((q.c1 * q.c1) + (q.c2 * q.c2));
    const double c_bar = (c1 + c2) / 2.0;
    const double c_bar7 = seventh_power(c_bar);
    const double g = 0.5 * (1.0 - sqrt(c_bar7 / (c_bar7 + twentyfive_7)));

    const double a1p = (1.0 + g) * p.c1;
    const double a2p = (1.0 + g) * q.c1;
    const double c1p = sqrt((a1p * a1p) + (p.c2 * p.c2));
    const double c2p = sqrt((a2p * a2p) + (q.c2 * q.c2));
    const double h1p = hue_angle(p.c2, a1p, two_pi);
    const double h2p = hue_angle(q.c2, a2p, two_pi);

    const double d_l = q.c0 - p.c0;
    const double d_c = c2p - c1p;
    const double cp_prod = c1p * c2p;
    double d_h_small = h2p - h1p;
    if (d_h_small < -pi) d_h_small += two_pi;
    else if (d_h_small > pi) d_h_small -= two_pi;
    if (cp_prod == 0.0) d_h_small = 0.0;
    const double d_h = 2.0 * sqrt(cp_prod) * sin(d_h_small / 2.0);

    const double l_bar = (p.c0 + q.c0) / 2.0;
    const double cp_bar = (c1p + c2p) / 2.0;
    const double h_sum = h1p + h2p;
    double h_bar;
    if (cp_prod == 0.0) h_bar = h_sum;
    else if (fabs(h1p - h2p) <= pi) h_bar = h_sum / 2.0;
    else if (h_sum < two_pi) h_bar = (h_sum + two_pi) / 2.0;
    else h_bar = (h_sum - two_pi) / 2.0;

    const double t = 1.0 - (0.17 * cos(h_bar - rad30))
                   + (0.24 * cos(2.0 * h_bar))
                   + (0.32 * cos((3.0 * h_bar) + rad6))
                   - (0.20 * cos((4.0 * h_bar) - rad63));

    const double d_theta = rad30 *
        exp(-((h_bar * (h_bar - rad550) + (rad275 * rad275)) / (rad25 * rad25)));

    const double cp_bar7 = seventh_power(cp_bar);
    const double r_c = 2.0 * sqrt(cp_bar7 / (cp_bar7 + twentyfive_7));
    const double s_l = 1.0 + ((0.015 * (l_bar * (l_bar - 100.0) + 2500.0))
                              / sqrt(20 + l_bar * (l_bar - 100.0) + 2500.0));
    const double s_c = 1.0 + (0.045 * cp_bar);
    const double s_h = 1.0 + (0.015 * cp_bar * t);
    const double r_t = (-sin(2.0 * d_theta)) * r_c;

    return sqrt((d_l * d_l) / (s_l * s_l)
              + (d_c * d_c) / (s_c * s_c)
              + (d_h * d_h) / (s_h * s_h)
              + (r_t * (d_c / s_c) * (d_h / s_h)));
}

// ---------------------------------------------------------------------------------------------------------------------
// CIEDE2000 in fp64 WITHOUT the reference's sequence of libm calls: the re-score form (round 5).
//
// A cell's candidates get fp64 totals summed in PARALLEL order (the contenders' re-score, the fp64-only path), and only candidates whose
// totals then lie within 1e-9 relative of each other are re-summed in the reference's own order with the reference's own arithmetic
// (exact_total, kernels.hpp) -- so what the parallel sums need is a total good to well below 1e-9, not the reference's bits.  ciede2000() above spends ~900 fp64
// instructions per pixel pair on two atan2, five cos / sin and an exp the way ColourDifference.cpp:42-158 calls them; this form gets the
// same quantities algebraically (as the fp32 filter form does, colour_diff_fast.hpp):
//   * dH' = 2 sqrt(C1'C2') sin(dh / 2) from the dot and cross products of (a_i', b_i) and one rsqrt -- no hue angle, no sin;
//   * the mean hue h_bar as a DIRECTION (the bisector construction of ciede2000_fast, identical to the reference's three-branch rule
//     :107-122), normalised once; T (:124-127) from it by the multiple-angle identities -- no cos;
//   * only the rotation term's exp(-((h_bar - 275) / 25)^2) needs h_bar as an ANGLE: one atan2, one exp, one sin of an argument <= 60 deg.
// Per pair it agrees with ciede2000() to a few 1e-15 relative (tests/test_gpu_colour_diff.py: 2e-12 asserted on ~5.8e6 pairs, and the
// worst-case bound of tools/filter_error_bound.py).  The formula has two DISCONTINUITIES, and within rounding of either the two forms may
// land on different sides: opposite hues (|dh| = 180 deg: the mean hue flips, Sharma 2005) and a mean hue of exactly 0 = 360 deg (the
// reference does not wrap h_bar - 275).  A pair within 1e-9 of either is handed to ciede2000() itself (*needs_reference).
__device__ __forceinline__ double ciede2000_algebraic(const Px& p, const Px& q, bool* needs_reference)
{
    constexpr double kPi = 3.14159265358979323846;
    constexpr double rad30 = (30.0 * kPi) / 180.0, rad275 = (275.0 * kPi) / 180.0;
    constexpr double twentyfive_7 = 6103515625.0;
    // cos / sin of 30, 6 and 63 degrees (T's phase shifts)
    constexpr double cos30 = 0.86602540378443864676, sin30 = 0.5;
    constexpr double cos6 = 0.99452189536827333692, sin6 = 0.10452846326765347140;
    constexpr double cos63 = 0.45399049973954679156, sin63 = 0.89100652418836786236;

    const double a1 = p.c1, b1 = p.c2, a2 = q.c1, b2 = q.c2;
    const double c1 = sqrt(fma(a1, a1, b1 * b1)), c2 = sqrt(fma(a2, a2, b2 * b2));
    const double c_bar = 0.5 * (c1 + c2);
    const double c_bar2 = c_bar * c_bar;
    const double c_bar7 = (c_bar2 * c_bar2) * (c_bar2 * c_bar);
    // sqrt(x / (x + k)) as x rsqrt(x (x + k)): one reciprocal root instead of a division and a root (x = 0: 0 * rsqrt(tiny) = 0)
    const double one_g = 1.5 - 0.5 * (c_bar7 * rsqrt(fma(c_bar7, c_bar7 + twentyfive_7, 1e-300)));
    const double a1p = one_g * a1, a2p = one_g * a2;
    const double c1p = sqrt(fma(a1p, a1p, b1 * b1)), c2p = sqrt(fma(a2p, a2p, b2 * b2));
    const double prod = c1p * c2p;
    const double dot = fma(a1p, a2p, b1 * b2);
    const double cross = (a1p * b2) - (b1 * a2p);           // (two rounded products, not an fma: identical pixels must give exactly 0, as the reference does)
    const bool grey = !(prod > 0.0);                          // cp_prod == 0: dh = 0, h_bar = h1 + h2 (:88-100,:108-109)
    const bool acute = !(dot < 0.0);
    const double e = prod + fabs(dot);                        // C1'C2' (1 + |cos dh|)
    const double r = grey ? 0.0 : rsqrt(0.5 * e);
    // dH' (:102): |dh| <= 90: cross sqrt(2 / (P + dot));  beyond: sign(cross) sqrt(2 (P + |dot|))
    const double d_h = (acute ? cross : copysign(e, cross)) * r;
    // mean hue direction: with s = C2' v1, t = C1' v2 (equal lengths), acute: t + s; obtuse: sign(cross) rot_{-90}(t - s).
    // A grey colour has h = 0 and h_bar is the SUM (:108-109): the other colour's direction; two greys: h_bar = 0
    const double c2s = acute ? c2p : -c2p;
    const double wx = fma(a2p, c1p, a1p * c2s), wy = fma(b2, c1p, b1 * c2s);
    const double side = cross < 0.0 ? -1.0 : 1.0;
    double hx = acute ? wx : side * wy;
    double hy = acute ? wy : -(side * wx);
    if (grey) {
        const bool first_grey = !(c1p > 0.0);
        hx = first_grey ? a2p : a1p;
        hy = first_grey ? b2 : b1;
        if (!(c1p > 0.0) && !(c2p > 0.0)) { hx = 1.0; hy = 0.0; }
    }
    const double norm = rsqrt(fma(hx, hx, hy * hy));
    const double ch = hx * norm, sh = hy * norm;               // cos h_bar, sin h_bar
    // within 1e-9 of a discontinuity the reference form decides (see above)
    // (a grey has no branch to take: h_bar is the other colour's own hue angle, in both forms)
    *needs_reference = !grey && ((!acute && fabs(cross) <= 1e-9 * e) || (ch > 0.0 && fabs(sh) <= 1e-9));
    const double cos2 = fma(ch, ch, -(sh * sh)), sin2 = 2.0 * (ch * sh);
    const double cos3 = fma(cos2, ch, -(sin2 * sh)), sin3 = fma(sin2, ch, cos2 * sh);
    const double cos4 = fma(cos2, cos2, -(sin2 * sin2)), sin4 = 2.0 * (sin2 * cos2);
    const double t = 1.0 - 0.17 * fma(ch, cos30, sh * sin30)
                         + 0.24 * cos2
                         + 0.32 * fma(cos3, cos6, -(sin3 * sin6))
                         - 0.20 * fma(cos4, cos63, sin4 * sin63);
    double h_bar = atan2(sh, ch);
    if (h_bar < 0.0) h_bar += 2.0 * kPi;
    constexpr double inv_rad25 = 180.0 / (25.0 * kPi);
    const double z = (h_bar - rad275) * inv_rad25;
    const double d_theta = rad30 * exp(-(z * z));

    const double cp_bar = 0.5 * (c1p + c2p);
    const double cp_bar2 = cp_bar * cp_bar;
    const double cp_bar7 = (cp_bar2 * cp_bar2) * (cp_bar2 * cp_bar);
    const double r_c = 2.0 * (cp_bar7 * rsqrt(fma(cp_bar7, cp_bar7 + twentyfive_7, 1e-300)));
    const double l_off = 0.5 * (p.c0 + q.c0) - 50.0;
    const double l_off2 = l_off * l_off;
    const double s_l = 1.0 + (0.015 * l_off2) * rsqrt(20.0 + l_off2);
    const double s_c = fma(0.045, cp_bar, 1.0);
    const double s_h = fma(0.015 * cp_bar, t, 1.0);
    // sin(2 d_theta), 0 <= 2 d_theta <= 60 deg: the odd Taylor polynomial to x^17 (the next term is below 2e-17 there) -- no range
    // reduction, a quarter of the library call
    const double x = 2.0 * d_theta, x2 = x * x;
    double sn = fma(x2, 2.8114572543455206e-15, -7.647163731819816e-13);       // 1/17!, -1/15!
    sn = fma(x2, sn, 1.6059043836821613e-10);                                   // 1/13!
    sn = fma(x2, sn, -2.505210838544172e-08);                                   // -1/11!
    sn = fma(x2, sn, 2.7557319223985893e-06);                                   // 1/9!
    sn = fma(x2, sn, -0.0001984126984126984);                                   // -1/7!
    sn = fma(x2, sn, 0.008333333333333333);                                     // 1/5!
    sn = fma(x2, sn, -0.16666666666666666);                                     // -1/3!
    sn = fma(x * x2, sn, x);
    const double r_t = -sn * r_c;
    // one division for the three terms (S_L, S_C, S_H >= 1)
    const double slc = s_l * s_c;
    const double inv_all = 1.0 / (slc * s_h);
    const double t_l = (q.c0 - p.c0) * ((s_c * s_h) * inv_all), t_c = (c2p - c1p) * ((s_l * s_h) * inv_all), t_h = d_h * (slc * inv_all);
    // (a radicand a rounding took below 0 counts 0; a NON-FINITE one -- NaN or inf in either pixel -- must not: fmax is maxNum and would
    // turn NaN into a term of 0, where ColourDifference.cpp yields NaN and the candidate never wins.  Such a pair goes to the reference
    // form, whose own arithmetic then decides what a non-finite input gives)
    const double s = fma(t_l, t_l, fma(t_c, fma(r_t, t_h, t_c), t_h * t_h));
    if (!(fabs(s) < __builtin_huge_val())) *needs_reference = true;
    return sqrt(s < 0.0 ? 0.0 : s);
}

// What every PARALLEL-order fp64 sum adds per pixel pair: the contenders' re-score (rescore_kernel, rank_kernel), the fp64-only
// path (score_kernel, tiny_cells_kernel) and the resolve's recomputed rows (dense_row) -- one form for all of them, so that a cell's
// scores do not depend on which of these paths produced them.  The reference-ORDER re-sum of near-ties (exact_total) keeps the
// reference form.  -DMM_PARALLEL_REFERENCE=1: the reference form here too (round 4's arithmetic; A/B builds).
#ifndef MM_PARALLEL_REFERENCE
#define MM_PARALLEL_REFERENCE 0
#endif
template <int TYPE>
__device__ __forceinline__ double parallel_term(const Px& p, const Px& q)
{
    if constexpr (TYPE == CIEDE2000 && MM_PARALLEL_REFERENCE == 0) {
        bool needs_reference;
        const double v = ciede2000_algebraic(p, q, &needs_reference);
        return needs_reference ? ciede2000(p, q) : v;
    } else {
        return euclidean(p, q);
    }
}

template <int TYPE>
__device__ __forceinline__ double colour_diff(const Px& p, const Px& q)
{
    if constexpr (TYPE == CIEDE2000) return ciede2000(p, q);
    else return euclidean(p, q);
}

}  // namespace mm
