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
//   * hue angles never materialise.  With u_i = (a_i', b_i) / C_i' the unit hue
//     vectors, cos(dh) = u1.u2 and sin(dh) = u1 x u2; half-angle identities give
//     sin(dh/2), cos(dh/2) with ONE rsq, hence dH' = 2 sqrt(C1'C2') sin(dh/2) and
//     the mean hue direction (cos, sin)(h1 + dh/2) -- exactly the reference's
//     mean-hue case analysis (:107-122), which is "h1 plus half the wrapped
//     difference" in all three branches;
//   * T (:124-127) and the rotation term's sin(2 d_theta) (:132-133,:145) depend on the
//     mean hue angle alone: one interpolated lookup in a 2048-entry LDS table indexed
//     by the "diamond angle" of the mean-hue vector (no inverse trig, one v_rcp_f32);
//   * the per-pixel chroma sqrt(a^2+b^2) is read from precomputed planes;
//   * sqrt(x/(x+k)) = x * rsq(x*(x+k)); divisions are v_rcp_f32.
// About 13 quarter-rate ops + ~100 full-rate ops per pixel pair, against ~900 fp64
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
    return fsqrt(fmaf(d0, d0, fmaf(d1, d1, d2 * d2)));
}

// (x/25)^7 / ((x/25)^7 + 1) under a square root, as x7 * rsq(x7 * (x7 + 1)).
__device__ __forceinline__ float sqrt_ratio7(float x)
{
    const float s = x * 0.04f;
    const float s2 = s * s, s4 = s2 * s2;
    const float z = s4 * s2 * s;
    return z * frsq(fmaxf(z * (z + 1.0f), 1e-36f));
}

// Hue table: T (ColourDifference.cpp:124-127) and sin(2*d_theta) (:132-133,:145) depend
// on the mean hue angle ALONE, so both are tabulated over the circle and interpolated
// linearly.  The index is the "diamond angle" of the mean-hue unit vector (hc, hs):
//   t = 2 - copysign(1 + hc / (|hc| + |hs|), hs)   in [0, 4),  monotone in the angle,
// which needs one v_rcp_f32 and no inverse trig.  Entry i holds {T, dT, G, dG} at
// t_i = 4 i / kHueLutSize with d* the difference to entry i+1 (kHueLutSize + 1 entries;
// the last repeats the first).  Built in fp64 on the host (engine.hip build_hue_lut).
// Interpolation error at 2048 entries: < 2e-5 on T, < 2e-5 on G.
constexpr int kHueLutSize = 2048;

// Returns the fp32 estimate; *needs_exact is set when the pair sits on the
// opposite-hue discontinuity and the caller must use the fp64 form instead.
// c1, c2: chroma sqrt(a^2 + b^2) of p and q, precomputed per pixel (stored planes).
// lut: the hue table in LDS.
__device__ __forceinline__ float ciede2000_fast(const PxF& p, const PxF& q, float c1, float c2,
                                                const float4* __restrict__ lut, bool* needs_exact)
{
    // G, a', C' (:50-59)
    const float one_g = 1.5f - 0.5f * sqrt_ratio7(0.5f * (c1 + c2));     // 1 + G
    const float a1 = one_g * p.c1, a2 = one_g * q.c1;
    const float m1 = fmaf(a1, a1, p.c2 * p.c2), m2 = fmaf(a2, a2, q.c2 * q.c2);
    const bool grey1 = m1 < 1e-30f, grey2 = m2 < 1e-30f;
    const float i1 = grey1 ? 0.0f : frsq(m1), i2 = grey2 ? 0.0f : frsq(m2);
    const float c1p = m1 * i1, c2p = m2 * i2;

    // unit hue vectors; a grey colour takes the other's hue (h' = 0 and h_bar =
    // h1 + h2 in the reference, :61-83,:108-109), two greys take hue 0
    float u1x = a1 * i1, u1y = p.c2 * i1, u2x = a2 * i2, u2y = q.c2 * i2;
    if (grey1) { u1x = grey2 ? 1.0f : u2x; u1y = grey2 ? 0.0f : u2y; }
    if (grey2) { u2x = u1x; u2y = u1y; }

    const float cosd = fmaf(u1x, u2x, u1y * u2y);          // cos(h2 - h1)
    const float sind = fmaf(u1x, u2y, -(u1y * u2x));       // sin(h2 - h1)
    *needs_exact = cosd < 0.0f && fabsf(sind) < 3e-5f;

    // half angle: sh = sin(dh/2) (signed), ch = cos(dh/2) >= 0, dh wrapped to (-pi, pi]
    const bool acute = cosd > 0.0f;
    const float e = acute ? 1.0f + cosd : 1.0f - cosd;     // in [1, 2]
    const float r = frsq(e + e);
    const float sh = acute ? sind * r : copysignf(e * r, sind);
    const float ch = acute ? e * r : fabsf(sind) * r;

    const float d_l = q.c0 - p.c0;
    const float d_c = c2p - c1p;
    const float d_h = 2.0f * fsqrt(c1p * c2p) * sh;         // :102

    // mean hue direction (:104-122) -> table index -> T and sin(2 d_theta)
    const float hc = fmaf(u1x, ch, -(u1y * sh));
    const float hs = fmaf(u1y, ch, u1x * sh);
    const float qd = hc * frcp(fmaxf(fabsf(hc) + fabsf(hs), 1e-20f));
    const float x = (2.0f - copysignf(1.0f + qd, hs)) * (float)(kHueLutSize / 4);
    const int idx = min(max((int)x, 0), kHueLutSize - 1);      // x >= 0: truncation = floor
    const float fr = x - (float)idx;                            // x == N (angle 2 pi) -> entry N-1, fr = 1
    const float4 en = lut[idx];
    const float t = fmaf(fr, en.y, en.x);
    const float s2 = fmaf(fr, en.w, en.z);

    const float cpb = 0.5f * (c1p + c2p);
    const float r_t = -s2 * 2.0f * sqrt_ratio7(cpb);        // :136,:145
    const float lb = 0.5f * (p.c0 + q.c0) - 50.0f;
    const float l2 = lb * lb;
    const float s_l = fmaf(0.015f * l2, frsq(20.0f + l2), 1.0f);
    const float s_c = fmaf(0.045f, cpb, 1.0f);
    const float s_h = fmaf(0.015f * cpb, t, 1.0f);

    const float t_l = d_l * frcp(s_l), t_c = d_c * frcp(s_c), t_h = d_h * frcp(s_h);
    const float xx = fmaf(t_l, t_l, fmaf(t_c, t_c, fmaf(t_h, t_h, r_t * t_c * t_h)));
    return fsqrt(fmaxf(xx, 0.0f));
}

__device__ __forceinline__ float chroma_of(float a, float b) { return fsqrt(fmaf(a, a, b * b)); }

}  // namespace mm
