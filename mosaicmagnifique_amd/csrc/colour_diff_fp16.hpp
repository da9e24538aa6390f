// colour_diff_fp16.hpp -- the PACKED fp16 form of CIEDE2000: the filter's optional pre-tier.
//
// Compiled only with -DMM_WITH_PRETIER=1 (launch.hpp).  The default build leaves the pre-tier out: on its certified band -- the
// worst-case bound of tools/filter_error_bound.py, 0.45 + 3.8e-2 dE per pixel pair -- it loses on every benchmarked configuration
// (config D 44.7 -> 48.8 ms per step, profiles/r05_pretier_band_ab.txt), and a tier that never runs should cost the dominant kernel
// neither source lines nor template instantiations.  What it bought on the OBSERVED band of round 4 (config D -6.3 %, 128-px cells
// -11.6 %) and why that band is a statistical guarantee, not a proof: DESIGN.md section 4, HISTORY.md rounds 4 and 5.
#pragma once

#include "colour_diff_fast.hpp"

namespace mm {

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
// (tools/experiments/archive/r4_sdwa_test.hip) -- a wait state between the two as well.)
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
