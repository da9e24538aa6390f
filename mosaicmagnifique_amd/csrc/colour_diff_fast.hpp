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

// ---------------------------------------------------------------------------------------------------------------------
// The PACKED fp16 form: the filter's pre-tier (filter.hip filter_kernel<.., PRE>).
//
// Two pixel pairs per lane-instruction (v_pk_*_f16).  It takes ONE kind of decision: "this candidate's sum over the
// groups visited so far, plus the bound of the others, already exceeds the cell's limit" -- with a limit widened by this
// form's own error band (filter.hip kPreAbs / kPreRel, measured by band_sweep_kernel modes 2 and 3 as the fp32 band is).
// Every candidate that survives it is scored again by the fp32 form above, which alone feeds the contender selection.
// A term may therefore be UNDER-estimated freely (terms are >= 0: a lower sum exits later, never wrongly), and that is
// what happens to everything awkward: pairs near the opposite-hue seam (|cross| < 2^-7 e: the fp16 branch choice may differ
// from the fp64 one), pairs with a chroma below 1 / 16 on either side (squares that underflow) and anything that came out NaN contribute 0.  What may NOT happen is an over-estimate beyond the band,
// so no intermediate may overflow for inputs inside the validated box (L in [0, 100], chroma <= 150):
//   operands are prepared (interleave_h_kernel): L / 2 - 25, a / 4, b / 4, C / 4 -- a' <= 57, products <= 4600, e <= 9200;
//   the 7th power of the mean chroma over 25 is clamped at 4.8^7 (the ratio is 1 - 9e-6 there) and enters as
//   rsq(1 + rcp(z)) -- z -> 0 gives rcp = inf, rsq(inf) = 0, the right limit -- instead of z rsq(z^2 + z), whose square
//   leaves fp16's range on either side; e carries + 2^-13 under its rsq (a chroma of 0.04: no hue to speak of);
//   tC and tH are formed before they are squared (tH <= 166 in the box, its square 27 600).
// The table index needs 11 bits and a fraction: index and interpolation stay in fp32 per pixel.
typedef _Float16 half1_t;
typedef _Float16 half2_t __attribute__((ext_vector_type(2)));

__device__ __forceinline__ half2_t h2_splat(float v) { return half2_t{(half1_t)v, (half1_t)v}; }
__device__ __forceinline__ unsigned int h2_bits(half2_t v) { return __builtin_bit_cast(unsigned int, v); }
__device__ __forceinline__ half2_t h2_from(unsigned int v) { return __builtin_bit_cast(half2_t, v); }
// A transcendental handles ONE half per instruction.  Written through the builtins the compiler computes the high half into
// the low word of a second register (SDWA source select) and spends a v_pack_b32_f16 to join the two -- 22 of them per four
// pixels.  The SDWA destination select does the join itself: the second instruction writes the HIGH word of the register the
// first one filled and preserves the low one.  (gfx940 / gfx950: a VALU write through dst_sel needs ONE wait state before a VALU reads
// that register -- the compiler inserts it for its own code, inline asm is not hazard-checked: the s_nop 0.  Found the hard way: without
// it the structured sweep returned inf and config G's golden failed.  And the SDWA instruction's PRESERVE is itself a read of the register
// the e32 instruction has just written: issued back to back it kept a stale low half in a third of the cases
// (tools/experiments/r4_sdwa_test.hip) -- a wait state between the two as well.)
#ifndef MM_PRE_ONE_RCP
#define MM_PRE_ONE_RCP 1             // ciede2000_h2: one reciprocal for t_L, t_C and t_H (0: two, as the fp32 form has them)
#endif
#ifndef MM_SDWA_GAP
#define MM_SDWA_GAP "s_nop 0\n\t"     // wait states between the low half's write and the SDWA instruction that PRESERVES it (see above)
#endif
// The same forwarding hazard exists on the way IN: should the compiler ever produce the operand with a dst_sel / op_sel-high write of
// its own in the instruction just before the block (today it is always a full-register v_pk_* result), the e32 instruction would read
// it one wait state early, and the hazard recognizer does not look into inline asm -- so the block opens with a wait state as well
// (MM_SDWA_LEAD; measured cost in HISTORY.md).  The encodings and wait states are gfx950's: any other target takes the builtins and
// lets the compiler join the halves (the v_pack_b32_f16 per pair this block exists to avoid).
#ifndef MM_SDWA_LEAD
#define MM_SDWA_LEAD "s_nop 0\n\t"
#endif
#if defined(__gfx950__) || !defined(__HIP_DEVICE_COMPILE__)
#define MM_H2_UNARY(NAME, OP, BUILTIN)                                                                                     \
    __device__ __forceinline__ half2_t NAME(half2_t v)                                                                     \
    {                                                                                                                      \
        unsigned int in = h2_bits(v), out;                                                                                 \
        asm(MM_SDWA_LEAD OP "_e32 %0, %1\n\t" MM_SDWA_GAP OP "_sdwa %0, %1 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:WORD_1\n\ts_nop 0" \
            : "=&v"(out) : "v"(in));                                                                                       \
        return h2_from(out);                                                                                               \
    }
#else
#define MM_H2_UNARY(NAME, OP, BUILTIN)                                                                                     \
    __device__ __forceinline__ half2_t NAME(half2_t v) { return half2_t{(half1_t)BUILTIN((__fp16)v.x), (half1_t)BUILTIN((__fp16)v.y)}; }
#endif
MM_H2_UNARY(h2_rsq, "v_rsq_f16", __builtin_amdgcn_rsqh)
MM_H2_UNARY(h2_rcp, "v_rcp_f16", __builtin_amdgcn_rcph)
MM_H2_UNARY(h2_sqrt, "v_sqrt_f16", __builtin_amdgcn_sqrth)
#undef MM_H2_UNARY
// the larger of two NON-NEGATIVE halfs, per half (unsigned order of the bit patterns; a NaN wins)
typedef unsigned short ushort2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ half2_t h2_umax(half2_t a, half2_t b)
{
    return __builtin_bit_cast(half2_t, __builtin_elementwise_max(__builtin_bit_cast(ushort2_t, a), __builtin_bit_cast(ushort2_t, b)));
}
// per-half select by a mask that holds 0xffff or 0 in each half
__device__ __forceinline__ half2_t h2_sel(unsigned int mask, half2_t a, half2_t b) { return h2_from((h2_bits(a) & mask) | (h2_bits(b) & ~mask)); }
// 0xffff in every half whose sign bit is set in `signs` (a value masked with 0x80008000)
__device__ __forceinline__ unsigned int h2_sign_mask(unsigned int signs) { const unsigned int t = signs >> 15; return (t << 16) - t; }

// sqrt(z / (z + 1)), z = s^7, for s = mean chroma / 25
#ifndef MM_PRE_CHROMA_FLOOR
#define MM_PRE_CHROMA_FLOOR 1        // ciede2000_h2: C' floored by the stored chroma (0: A/B builds only -- unsound for chroma 1e-5 .. 1e-3 beside saturated colours)
#endif
#ifndef MM_PRE_RATIO_ONE_RSQ
#define MM_PRE_RATIO_ONE_RSQ 1       // one transcendental per ratio instead of two (0: rsq(1 + rcp(z)))
#endif
__device__ __forceinline__ half2_t h2_ratio7(half2_t s)
{
#if MM_PRE_RATIO_ONE_RSQ
    // z / (z + 1) = y^2 / (y^2 + y / 16) for y = z / 16, so the ratio is y rsq(y (y + 1/16)): per-half transcendentals cost two
    // instructions a pair and wait states around them, packed multiplies one.  s is clamped at 3.26 (y <= 245, y^2 inside the
    // halfs' range; the ratio is within 1.3e-4 of 1 from there on), and 2^-24 -- the smallest subnormal half -- keeps the root's operand
    // above zero where y underflows (s < 0.1, where the ratio is below 3e-4: it comes out as 0)
    s = __builtin_elementwise_min(s, h2_splat(3.26f));
    const half2_t s2 = s * s;
    const half2_t y = ((s2 * s2) * s2) * (s * h2_splat(1.0f / 16.0f));
    return y * h2_rsq(y * (y + h2_splat(1.0f / 16.0f)) + h2_splat(0x1p-24f));
#else
    s = __builtin_elementwise_min(s, h2_splat(4.8f));
    const half2_t s2 = s * s;
    const half2_t z = ((s2 * s2) * s2) * s;                         // <= 58 700
    return h2_rsq(h2_splat(1.0f) + h2_rcp(z));
#endif
}

constexpr float kHalfOperandScale = 0.25f;     // a and b are stored / 4
constexpr float kHalfChromaScale = 12.5f;      // ... and the chroma plane, held as C / 50 in the fp32 copies, as C / 4 as well

__device__ __forceinline__ half2_t ciede2000_h2(half2_t pl, half2_t pa, half2_t pb, half2_t pc, half2_t ql, half2_t qa, half2_t qb, half2_t qc,
                                                const HueEntry* __restrict__ lut, unsigned int* seam_out = nullptr)
{
    const half2_t one_g = h2_splat(1.5f) - h2_splat(0.5f) * h2_ratio7((pc + qc) * h2_splat(0.08f));     // mean C / 25 = (C1 / 4 + C2 / 4) x 4 / 2 / 25
    const half2_t a1 = one_g * pa, a2 = one_g * qa;
    // C' >= C (G >= 1): the stored chroma is a floor for C'.  It is what C' comes out as when the squares underflow -- a pixel with a chroma
    // of 1e-5 .. 1e-3 (greys as a float conversion leaves them) -- : with C1' = 0 beside a saturated C2' the cross product below is still
    // right but `e` is not, and dH' came out six times too large (+1.3 on a pair of 33.7: what tests/test_gpu_filter.py's maximiser
    // found first when it was pointed at this form)
#if MM_PRE_CHROMA_FLOOR
    // (non-negative halfs order like their bit patterns: one v_pk_max_u16 each -- the floating-point maximum quiets both operands first, three instructions)
    const half2_t c1p = h2_umax(h2_sqrt(a1 * a1 + pb * pb), pc), c2p = h2_umax(h2_sqrt(a2 * a2 + qb * qb), qc);
#else
    const half2_t c1p = h2_sqrt(a1 * a1 + pb * pb), c2p = h2_sqrt(a2 * a2 + qb * qb);
#endif
    const half2_t dot = a1 * a2 + pb * qb;
    const half2_t cross = a1 * qb - pb * a2;
    const unsigned int obtuse = h2_sign_mask(h2_bits(dot) & 0x80008000u);
    const unsigned int sgn = h2_bits(cross) & 0x80008000u;
    const half2_t abs_cross = h2_from(h2_bits(cross) & 0x7fff7fffu);
    const half2_t e = (c1p * c2p + h2_splat(0x1p-13f)) + h2_from(h2_bits(dot) & 0x7fff7fffu);
    // near the seam (obtuse, |cross| < e / 128, i.e. within ~8e-3 rad of opposite hues -- an order above fp16's noise on cross / e)
    // the pair counts 0: `seam` = 0xffff per half there.  And so does a pair with a stored chroma below 2^-6 (C < 1 / 16) on either side:
    // the squares of a / 4 and b / 4 are subnormal halfs from there down, C' and the hue direction lose their relative accuracy, and a
    // grey's hue terms beside a saturated colour are small but not zero -- the zone where the worst-case bound of this form
    // (tools/filter_error_bound.py) exceeded any band worth having.  Non-negative halfs order like their bit patterns: the sign bit of
    // the 16-bit difference to 0x2400 (a NaN -- an inactive pixel -- lies above it and comes out as 0 further down anyway).
    const unsigned int tiny = h2_bits(__builtin_bit_cast(half2_t, __builtin_bit_cast(ushort2_t, pc) - ushort2_t{0x2400, 0x2400})) |
                              h2_bits(__builtin_bit_cast(half2_t, __builtin_bit_cast(ushort2_t, qc) - ushort2_t{0x2400, 0x2400}));
    const unsigned int seam = h2_sign_mask(((h2_bits(abs_cross - e * h2_splat(1.0f / 128.0f)) & h2_bits(dot)) | tiny) & 0x80008000u);
    const half2_t r = h2_rsq(h2_splat(0.5f) * e);
    const half2_t d_h = h2_sel(obtuse, h2_from(h2_bits(e) | sgn), cross) * r;
    const half2_t c2s = h2_from(h2_bits(c2p) ^ (obtuse & 0x80008000u));
    const half2_t wx = a2 * c1p + a1 * c2s, wy = qb * c1p + pb * c2s;
    const half2_t hx = h2_sel(obtuse, h2_from(h2_bits(wy) ^ sgn), wx);
    const half2_t hy = h2_sel(obtuse, h2_from(h2_bits(wx) ^ sgn ^ 0x80008000u), wy);
    float tt[2], ss[2];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const float fx = (float)(k ? hx.y : hx.x), fy = (float)(k ? hy.y : hy.x);
        const float qd = fx * frcp(fabsf(fx) + fabsf(fy) + 1e-30f);
        const float x = fmaf(-((float)(kHueLutSize / 4) - 0x1p-13f), copysignf(1.0f + qd, fy), (float)(kHueLutSize / 2));
        const int idx = (int)x;
        const float fr = __builtin_amdgcn_fractf(x);
        const HueEntry e0 = lut[idx], e1 = lut[idx + 1];
        tt[k] = fmaf(fr, e1.x - e0.x, e0.x);
        ss[k] = fmaf(fr, e1.y - e0.y, e0.y);
    }
    const half2_t t4 = half2_t{(half1_t)(4.0f * tt[0]), (half1_t)(4.0f * tt[1])};      // 0.0075 T x 4: the chroma sums are carried / 4
    const half2_t s2 = half2_t{(half1_t)ss[0], (half1_t)ss[1]};
    const half2_t csum = c1p + c2p;                                  // (C1' + C2') / 4
    const half2_t r_t = s2 * h2_ratio7(csum * h2_splat(0.08f));     // mean C' / 25 = csum x 4 / 2 / 25
    const half2_t lb = pl + ql;
    const half2_t u_l = lb * lb + h2_splat(20.0f);
    const half2_t s_l = (h2_splat(0.0075f) * u_l - h2_splat(0.15f)) * h2_rsq(u_l) + h2_splat(0.5f);      // S_L / 2
    const half2_t s_c = h2_splat(0.09f) * csum + h2_splat(1.0f);     // 1 + 0.045 mean C'
    const half2_t s_h = csum * t4 + h2_splat(1.0f);                 // 1 + 0.015 mean C' T
#if MM_PRE_ONE_RCP
    // one reciprocal for all three terms: per-half transcendentals cost two instructions a pair, packed multiplies one
    const half2_t big_d = s_c * s_h;                                // <= 72
    const half2_t inv_all = h2_rcp(big_d * s_l);                     // S_L / 2 in [0.5, 0.88]: the product stays within [0.5, 64]
    const half2_t inv_d = (inv_all * s_l) * h2_splat(4.0f);          // (the differences below are carried / 4)
    const half2_t t_l = (ql - pl) * (inv_all * big_d);
#else
    const half2_t inv_d = h2_rcp(s_c * s_h) * h2_splat(4.0f);       // (the differences below are carried / 4)
    const half2_t t_l = (ql - pl) * h2_rcp(s_l);
#endif
    const half2_t t_c = ((c2p - c1p) * s_h) * inv_d;
    const half2_t t_h = (d_h * s_c) * inv_d;
    const half2_t xx = t_l * t_l + (t_c * (r_t * t_h + t_c) + t_h * t_h);
    // max(NaN, 0) = 0: an inactive pixel (NaN in the cell's operands), or anything else that went wrong, adds nothing
    const half2_t v = h2_sqrt(__builtin_elementwise_max(xx, h2_splat(0.0f)));
    if (seam_out) *seam_out = seam;
    return h2_from(h2_bits(v) & ~seam);
}

}  // namespace mm
