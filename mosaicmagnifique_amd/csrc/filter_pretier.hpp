// filter_pretier.hpp -- the packed-fp16 pre-tier of filter_kernel<.., PRE> (compiled only with -DMM_WITH_PRETIER=1, launch.hpp).
//
// A FRAGMENT of filter.hip: included inside namespace mm::{anonymous} after the wave-sum helpers it uses
// (wave_sum_exceeds, wave_sum_all).  Nothing here is part of the default build.
//
// The packed-fp16 pre-tier (colour_diff_fp16.hpp ciede2000_h2).  A candidate first walks its groups in the fp16 form,
// two pixel pairs per lane-instruction, from half copies of the cell and the library (`cells` and `lib_chroma`, which the
// interleaved fp32 form leaves unused, carry their addresses), with the exit test against the limit widened by the fp16
// form's own band; only a candidate that survives ALL its groups there is walked again by the fp32 form, whose score is the
// one the contender selection reads.  Dropped candidates get +inf as before.  Sound because every term is >= 0 and the fp16
// sum under-estimates by at most its band: "fp16 sum of the visited groups + bound of the others > limit + band" implies the
// same for the true sum against the limit.
// Error band of the packed-fp16 pre-tier per pixel pair (launch.hpp kPreBandAbs / kPreBandRel): fp16 form - fp64 form <= abs + rel * value for operands
// inside the validated box, with the operands rounded to halfs as interleave_h_kernel stores them; the pre-tier's accumulation is fp32.
// Round 5: the band is no longer sized on OBSERVED errors.  tools/filter_error_bound.py carries, through the form's own sequence of
// operations, a bound on what the half-precision computation can be off by with EVERY rounding at its worst (u = 2^-11 per packed
// operation, one ulp per transcendental, subnormal halfs at their absolute step, the hue table's index moved by the direction's
// error) -- a smooth function of the pair, unlike an observed error, which depends on mantissa bits and let round 4's sweeps step over
// what the maximiser then found.  Its frontier over thirteen families x 1.5e6 pairs (the form's weak zones, the colours of u8 images):
// 0.40 + 3.3e-2 v, 0.45 + 3.1e-2 v, 0.50 + 2.9e-2 v; shipped with a tenth of room on either part.  Observed errors sit at a tenth of it
// (round 4's band, twice the worst observed pair, was 0.29 + 8e-3 v -- kPreBandAbsObserved / kPreBandRelObserved, option "pretier_band" 1:
// a third of the worst-case bound, i.e. a band that trusts the roundings not to conspire; config D ran 4 % faster on it).
#ifndef MM_PRE_PREFETCH
#define MM_PRE_PREFETCH 1            // the pre-tier fetches the next group's operands while it computes this one (0: A/B builds)
#endif

// The fp16 walk of one candidate by one wave.  ch / lh: the cell's and the library image's half copies ([group][plane][256] halfs --
// a group is 2 KB, a plane 512 B, a lane's four positions 8 B).  Returns true when the walk dropped the candidate; otherwise
// pre_sum holds the lane's partial fp16-form sum (for the canary) and the fp32 walk follows.  visited counts the groups walked.
// True sum of the visited pairs >= (acc16 - abs x pairs visited) / (1 + rel): the exit test compares acc16 + rest with
// limit (1 + rel) + abs x (positions visited so far) -- the absolute part grows with the walk instead of being charged for the
// whole cell up front.
__device__ __forceinline__ bool pretier_walk(const char* const ch, const char* const lh, int lane, int groups, int check_every,
                                             bool use_lb, float rest, bool by_group, int lane_shift, bool ordered, int order,
                                             float limit, float pre_abs, float pre_rel, const HueEntry* lut, int& visited, float& pre_sum)
{
    const float limit_rel = limit * (1.0f + pre_rel);
    float acc16 = 0.0f;
    bool out16 = false;
    // the operands of the NEXT group are fetched while this one is computed: a group of the fp16 form is ~280 instructions
    // (the fp32 one ~430) and its sixteen registers of operands leave room for a second set inside the kernel's 96
    struct HalfGroup { uint2 p0, p1, p2, pc, q0, q1, q2, qc; };
    auto fetch = [&](int j) {
        const unsigned int off = (unsigned int)j * 2048u + (unsigned int)lane * 8u;
        HalfGroup g;
        g.p0 = *reinterpret_cast<const uint2*>(ch + off); g.p1 = *reinterpret_cast<const uint2*>(ch + off + 512u);
        g.p2 = *reinterpret_cast<const uint2*>(ch + off + 1024u); g.pc = *reinterpret_cast<const uint2*>(ch + off + 1536u);
        g.q0 = *reinterpret_cast<const uint2*>(lh + off); g.q1 = *reinterpret_cast<const uint2*>(lh + off + 512u);
        g.q2 = *reinterpret_cast<const uint2*>(lh + off + 1024u); g.qc = *reinterpret_cast<const uint2*>(lh + off + 1536u);
        return g;
    };
    HalfGroup cur = fetch(ordered ? __builtin_amdgcn_readlane(order, 0) : 0);
    for (int step = 0; step < groups; ++step) {
        if (step != 0 && step % check_every == 0) {
            const float rest_j = use_lb ? __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, rest), by_group ? step : (4 * step) >> lane_shift)) : 0.0f;
            if (wave_sum_exceeds(acc16, rest_j, fmaf(pre_abs * (float)(4 * kWave), (float)step, limit_rel))) { out16 = true; break; }
        }
#if MM_PRE_PREFETCH
        const int nxt = step + 1 < groups ? step + 1 : step;         // (the last step fetches its own group again: no branch)
        const HalfGroup ahead = fetch(ordered ? __builtin_amdgcn_readlane(order, nxt) : nxt);
#else
        if (step != 0) cur = fetch(ordered ? __builtin_amdgcn_readlane(order, step) : step);
#endif
        const half2_t va = ciede2000_h2(h2_from(cur.p0.x), h2_from(cur.p1.x), h2_from(cur.p2.x), h2_from(cur.pc.x),
                                        h2_from(cur.q0.x), h2_from(cur.q1.x), h2_from(cur.q2.x), h2_from(cur.qc.x), lut);
        const half2_t vb = ciede2000_h2(h2_from(cur.p0.y), h2_from(cur.p1.y), h2_from(cur.p2.y), h2_from(cur.pc.y),
                                        h2_from(cur.q0.y), h2_from(cur.q1.y), h2_from(cur.q2.y), h2_from(cur.qc.y), lut);
        acc16 += ((float)va.x + (float)va.y) + ((float)vb.x + (float)vb.y);
        visited += 1;
#if MM_PRE_PREFETCH
        cur = ahead;
#endif
    }
    if (!out16 && wave_sum_exceeds(acc16, 0.0f, fmaf(pre_abs * (float)(4 * kWave), (float)groups, limit_rel))) out16 = true;
    if (!out16) pre_sum = acc16;        // (a survivor goes on to the fp32 form)
    return out16;
}

// Canary of the pre-tier: a candidate that walked ALL its groups in both forms has two sums of the same terms (the fp16 one
// without its seam pairs, which can only make it smaller).  An fp16 sum ABOVE the fp32 one by more than the band's worth says
// the fp16 form over-estimates here -- what its early exits must never do -- and voids the launch: a sticky bit in the
// counter block's void word (an OR: it cannot wrap, however many candidates trip it), which sends the whole batch down
// the fp64-only path (and shows in mm_timing.voided_launches).  acc: the fp32 form's wave-wide sum.
__device__ __forceinline__ void pretier_canary(float pre_sum, float acc, int lane, int groups, float pre_abs, float pre_rel,
                                               unsigned int* edge_count)
{
    const float sum16 = wave_sum_all(pre_sum);
    // (the band in force; the diagnostics' negative bands -- what the audit is shown to catch -- by their size)
    if (lane == 0 && sum16 > fmaf(acc, 1.0f + fabsf(pre_rel), fabsf(pre_abs) * (float)(groups * 4 * kWave)))
        atomicOr(edge_count + kFilterVoidWord, kVoidByCanary);
}
