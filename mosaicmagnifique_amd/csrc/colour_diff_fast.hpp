// colour_diff_fast.hpp -- fp32 FILTER forms of the colour differences (gfx950).
//
// These never decide a result on their own.  The filter pass scores every
// (cell, candidate) with them, keeps each cell's candidates whose filter score is
// within a proven-safe band of the K-th lowest, and the fp64 reference-order
// forms of colour_diff.hpp then re-score exactly those (engine.hip, DESIGN.md
// "Two-tier scoring").  What matters here is (a) a bounded error against the
// fp64 form and (b) instruction count: CIEDE2000 is VALU-bound, not HBM-bound.
//
// CIEDE2000 (ColourDifference.cpp:42-158) restated without atan2 / sin / cos calls:
//   * hue angles never materialise: dH' = 2 sqrt(C1'C2') sin(dh/2) and the mean hue
//     direction follow from the dot and cross products of (a_i', b_i) with ONE rsq --
//     exactly the reference's mean-hue case analysis (:107-122), which is "h1 plus
//     half the wrapped difference" in all three branches (see ciede2000_fast);
//   * T (:124-127) and the rotation term's sin(2 d_theta) (:132-133,:145) depend on the
//     mean hue angle alone: one interpolated lookup in a 2048-entry LDS table indexed
//     by the "diamond angle" of the mean-hue vector (no inverse trig, one v_rcp_f32);
//   * the per-pixel chroma sqrt(a^2+b^2) is read from precomputed planes;
//   * sqrt(x/(x+k)) = x * rsq(x*(x+k)); divisions are v_rcp_f32.
// 10 quarter-rate ops + 74 full-rate ops per pixel pair (ISA of filter_kernel: 344 VALU per four pairs, 382 before round 3's last pass), against ~900 fp64
// instructions for the reference form.
//
// The formula is DISCONTINUOUS where the two hues are opposite (|dh| = 180 deg: the
// mean hue flips by 180 deg, Sharma 2005 sec. 4).  Within 3e-5 rad of that edge the
// fp32 branch choice may differ from the fp64 one, so such pairs (about 1e-5 of all
// pairs) are evaluated with the fp64 reference form instead: `needs_exact`.
#pragma once

#include <hip/hip_runtime.h>

namespace mm {

struct PxF { float c0, c1, c2; };

__device__ __forceinline__ float frcp(float x) { return __builtin_amdgcn_rcpf(x); }
__device__ __forceinline__ float frsq(float x) { return __builtin_amdgcn_rsqf(x); }
__device__ __forceinline__ float fsqrt(float x) { return __builtin_amdgcn_sqrtf(x); }
__device__ __forceinline__ float fexp2(float x) { return __builtin_amdgcn_exp2f(x); }

__device__ __forceinline__ float euclidean_fast(const PxF& p, const PxF& q)
{
    const float d0 = p.c0 - q.c0, d1 = p.c1 - q.c1, d2 = p.c2 - q.c2;
    return fsqrt(fmaxf(fmaf(d0, d0, fmaf(d1, d1, d2 * d2)), 0.0f));      // max(NaN, 0) = 0: inactive pixel
}

// (x/25)^7 / ((x/25)^7 + 1) under a square root, as x7 * rsq(x7 * (x7 + 1)).
__device__ __forceinline__ float sqrt_ratio7_scaled(float s);
__device__ __forceinline__ float sqrt_ratio7(float x) { return sqrt_ratio7_scaled(x * 0.04f); }
// The same for a MEAN given as the sum of its two terms: (x1 + x2) / 2 / 25 = (x1 + x2) * 0.02, one multiply.
__device__ __forceinline__ float sqrt_ratio7_of_sum(float x_sum) { return sqrt_ratio7_scaled(x_sum * 0.02f); }
__device__ __forceinline__ float sqrt_ratio7_scaled(float s)
{
    const float s2 = s * s, s4 = s2 * s2;
    const float z = fmaf(s4 * s2, s, 1e-30f);       // > 0: no 0 * inf below
    return z * frsq(fmaf(z, z, z));
}

// Hue table: T (ColourDifference.cpp:124-127) and sin(2*d_theta) (:132-133,:145) depend
// on the mean hue angle ALONE, so both are tabulated over the circle and interpolated
// linearly.  The index is the "diamond angle" of the mean-hue unit vector (hc, hs):
//   t = 2 - copysign(1 + hc / (|hc| + |hs|), hs)   in [0, 4),  monotone in the angle,
// which needs one v_rcp_f32 and no inverse trig.  Entry i holds {0.0075 T, G} (S_H = 1 + 0.015 C T with C = csum / 2 is then one fma), G = -2 sin(2 d_theta)
// (so that R_T = G R_C / 2 is one multiply, :136,:145), at t_i = 4 i / kHueLutSize; a lookup reads
// entries i and i + 1 (one ds_read2_b64) and interpolates between them, so the table has
// kHueLutSize + 2 entries: entry N repeats entry 0 and entry N + 1 entry 1.  8 bytes per entry --
// 16 KB of LDS, not the 32 KB of a {value, slope} table: what lets six filter workgroups share a
// CU's LDS.  Built in fp64 on the host (engine.hip build_hue_lut).
// Interpolation error at 2048 entries: < 2e-5 on T, < 2e-5 on G.
constexpr int kHueLutSize = 2048;
constexpr int kHueLutEntries = kHueLutSize + 2;
using HueEntry = float2;

// Returns the fp32 estimate; *needs_exact is set when the pair sits on the
// opposite-hue discontinuity and the caller must use the fp64 form instead.
// c1, c2: chroma sqrt(a^2 + b^2) of p and q, precomputed per pixel (stored planes).
// lut: the hue table in LDS.
//
// Hue terms without unit vectors.  With v_i = (a_i', b_i), C_i = |v_i|, P = C1 C2,
// dot = v1.v2 = P cos(dh), cross = v1 x v2 = P sin(dh):
//   dH' = 2 sqrt(P) sin(dh/2)  (:102)  and  sin^2(dh/2) = (1 - cos dh)/2, so
//     |dh| <= 90:  dH'^2 = 2 (P - dot) = 2 cross^2 / (P + dot)      (no cancellation)
//     |dh| >  90:  dH'^2 = 2 (P + |dot|)
//   i.e. with e = P + |dot|, r = rsq(e / 2) = 2 rsq(2 e):  dH' = cross r   or   sign(cross) e r.
//   mean hue direction (:104-122, "h1 plus half the wrapped difference"), any positive
//   scale (the table index is scale-free): with s = C2 v1, t = C1 v2 (equal lengths)
//     |dh| <= 90:  s + t                       (bisector, length >= sqrt(2) P)
//     |dh| >  90:  sign(cross) rot_{-90}(t - s)  (same bisector, length >= sqrt(2) P)
// A grey colour (C = 0: h = 0, dh' = 0, h_bar = h1 + h2 in the reference, :61-83,:108-109)
// needs no case of its own: dot = cross = P = 0 gives dH' = 0 and the direction of the other
// colour (C' is never below 1e-18); two greys give direction (0, 0),
// where T and R_T are multiplied by a zero mean chroma.
//
// PREP = true: the operands come from the filter kernel's interleaved copies (filter.hip interleave_kernel), which
// hold channel 0 as L / 2 - 25 and the chroma as C / 50 -- the mean lightness offset (:138) is then one addition and
// the mean chroma over 25 (:50-53) needs no multiply; S_L is carried as S_L / 2 against the halved lightness difference.
template <bool PREP = false>
__device__ __forceinline__ float ciede2000_fast(const PxF& p, const PxF& q, float c1, float c2,
                                                const HueEntry* __restrict__ lut, bool* needs_exact)
{
    // G, a', C' (:50-59)
    const float one_g = 1.5f - 0.5f * (PREP ? sqrt_ratio7_scaled(c1 + c2) : sqrt_ratio7_of_sum(c1 + c2));       // 1 + G
    const float a1 = one_g * p.c1, a2 = one_g * q.c1;
    const float b1 = p.c2, b2 = q.c2;
    // + 1e-36: C' >= 1e-18 > 0 (a grey colour keeps a direction below, and e >= P stays positive under its rsq)
    const float m1 = fmaf(a1, a1, fmaf(b1, b1, 1e-36f)), m2 = fmaf(a2, a2, fmaf(b2, b2, 1e-36f));
#ifdef MM_CHROMA_RSQ                                      // (the form before round 3's last pass, kept for A/B builds)
    const float c1p = m1 * frsq(m1), c2p = m2 * frsq(m2);
#else
    const float c1p = fsqrt(m1), c2p = fsqrt(m2);            // one quarter-rate instruction each; m * rsq(m) cost a multiply more
#endif

    const float dot = fmaf(a1, a2, b1 * b2);
    const float cross = fmaf(a1, b2, -(b1 * a2));
    // (a comparison, not dot's sign bit: a grey colour -- a = b = 0 -- may give dot = -0.0 and must take the acute forms, which then
    // yield the other colour's direction)
    const bool acute = !(dot < 0.0f);
    const float e = fmaf(c1p, c2p, fabsf(dot));                             // P + |dot|, P = C1' C2'
    // within 3e-5 rad of opposite hues |cross| < 3e-5 P; there |dot| = P up to 1e-9, i.e. e = 2 P: the test reads e, and P itself
    // is never formed
    *needs_exact = !acute && fabsf(cross) < 1.50001e-5f * e;

    const float r = frsq(0.5f * e);                                         // = 2 rsq(2 e);  e >= P >= 1e-36
    // sign transplants as plain bit operations on one shared mask (full-rate, where v_bfi_b32 is not)
    const unsigned int sgn = __float_as_uint(cross) & 0x80000000u;
    // (selects, no divergent branch; the arm is chosen BEFORE the one multiply by r: e > 0 takes cross's sign as a bit operation)
    const float d_h = (acute ? cross : __uint_as_float(__float_as_uint(e) | sgn)) * r;     // dH'  (:102): 2 cross rsq(2e) | sign 2e rsq(2e)

    // mean hue direction -> table index -> T and sin(2 d_theta)
    // with s = C2' v1, t = C1' v2: w = t + s (acute) or t - s (obtuse) -- s takes its sign through C2', t rides in the fma
    const float c2s = acute ? c2p : -c2p;
    // (+ 1e-30 inside the first product's fma: w vanishes only when both colours are grey -- s = 0 needs v1 = 0, t = 0 needs v2 = 0,
    // and s = -+t with both non-zero contradicts the branch dot's sign chose -- and then the table index below must not see 0 / 0)
    const float wx = fmaf(a2, c1p, fmaf(a1, c2s, 1e-30f)), wy = fmaf(b2, c1p, b1 * c2s);
    // acute: w; obtuse: sign(cross) rot_{-90}(w) = sign(cross) (wy, -wx)
    const float hx = acute ? wx : __uint_as_float(__float_as_uint(wy) ^ sgn);
    const float hy = acute ? wy : __uint_as_float(__float_as_uint(wx) ^ sgn ^ 0x80000000u);
    const float qd = hx * frcp(fabsf(hx) + fabsf(hy));                      // |qd| <= 1 + 1 ulp (v_rcp_f32: 1 ulp); |hx| + |hy| > 0 (above)
    // (2 - c) N/4 with the factor N/4 shortened by 2^-13: even c = +-(2 + 2 ulp) keeps x inside [0, N], where
    // truncation = floor and fract = x - floor(x) agree (2.4e-4 of a table step at most; the table's own error is
    // 2e-5 of a value).  Entry N repeats entry 0 and N + 1 entry 1: no clamp.  NaN (an inactive pixel) converts to 0.
    const float x = fmaf(-((float)(kHueLutSize / 4) - 0x1p-13f), copysignf(1.0f + qd, hy), (float)(kHueLutSize / 2));
    const int idx = (int)x;
    const float fr = __builtin_amdgcn_fractf(x);                // x - floor(x), one instruction
    const HueEntry e0 = lut[idx], e1 = lut[idx + 1];
    const float t = fmaf(fr, e1.x - e0.x, e0.x);
    const float s2 = fmaf(fr, e1.y - e0.y, e0.y);

    const float csum = c1p + c2p;                           // 2 * mean C'
    const float r_t = s2 * sqrt_ratio7_of_sum(csum);            // :136,:145; the table holds -2 sin(2 d_theta)
    const float lb = PREP ? p.c0 + q.c0 : 0.5f * (p.c0 + q.c0) - 50.0f;
    // with u = 20 + lb^2 (one fma): 0.0075 lb^2 = 0.0075 u - 0.15 (one fma; the cancellation near lb = 0 costs 1e-8 of an S_L of 0.5)
    const float u_l = fmaf(lb, lb, 20.0f);
    const float s_l = PREP ? fmaf(fmaf(0.0075f, u_l, -0.15f), frsq(u_l), 0.5f)      // S_L / 2: d_l below is half the difference
                           : fmaf(fmaf(0.015f, u_l, -0.3f), frsq(u_l), 1.0f);
    const float s_c = fmaf(0.0225f, csum, 1.0f);
    const float s_h = fmaf(csum, t, 1.0f);                  // the table holds 0.0075 T: S_H = 1 + 0.015 (csum / 2) T

    // (:153-157) with ONE reciprocal for the chroma and hue terms:
    //   tC = dC/S_C = A/D, tH = dH/S_H = B/D,  D = S_C S_H, A = dC S_H, B = dH S_C
    const float d_l = q.c0 - p.c0;
    const float d_c = c2p - c1p;
    const float big_d = s_c * s_h;
    const float ta = d_c * s_h;
    const float tb = d_h * s_c;
    const float num = fmaf(ta, fmaf(r_t, tb, ta), tb * tb);        // ta^2 + tb^2 + R_T ta tb
    // (one reciprocal for all three terms -- rcp(S_L S_C S_H), three multiplies more -- measured 0.5 % slower: a quarter-rate
    // instruction costs what three full-rate ones do)
    const float inv_d = frcp(big_d);
    const float t_l = d_l * frcp(s_l);
    const float xx = fmaf(t_l, t_l, num * (inv_d * inv_d));
    return fsqrt(fmaxf(xx, 0.0f));          // also max(NaN, 0) = 0: an inactive pixel (c0 = NaN) adds nothing
}

__device__ __forceinline__ float chroma_of(float a, float b) { return fsqrt(fmaf(a, a, b * b)); }

}  // namespace mm
