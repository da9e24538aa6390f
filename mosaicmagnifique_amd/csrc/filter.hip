// filter.hip -- tier 1 of the two-tier scoring: fp32 filter scores of every
// (cell, candidate).  Built WITH fp contraction (fused multiply-add is wanted
// here); the fp64 reference-order kernels live in engine.hip, built without.
//
// Nothing in this translation unit decides a best fit: see colour_diff_fast.hpp.
#include <hip/hip_runtime.h>

#include <mutex>
#include <stdint.h>

#include <algorithm>
#include <cmath>

#include "colour_diff.hpp"
#include "colour_diff_fast.hpp"
#include "launch.hpp"
#if MM_WITH_PRETIER
#include "colour_diff_fp16.hpp"
#endif

namespace mm {

namespace {

constexpr int kBlock = 256;
constexpr int kWave = 64;

// Waves per SIMD the dominant kernel is compiled for: 5 -> at most 96 VGPRs (the 16 KB hue table and
// the block statistics leave room for five workgroups per CU).  Overridable for experiments.
#ifndef MM_FILTER_WAVES
#define MM_FILTER_WAVES 5
#endif
// ... and seed_score_kernel, which carries the fp64 form inline for pairs on the hue discontinuity: 3 -> 167 VGPRs,
// no spills (left alone the compiler takes 171 = 2 waves; at 4 and 5 the pixel loop spills: 1.78 / 1.41 / 2.06 / 3.57 ms
// for config D's seeds at 2 / 3 / 4 / 5).
#ifndef MM_SEED_WAVES
#define MM_SEED_WAVES 3
#endif
#ifndef MM_EUCLID_PREFETCH
#define MM_EUCLID_PREFETCH 0
#endif
#ifndef MM_EUCLID_WAVES
#define MM_EUCLID_WAVES 8
#endif

__device__ __forceinline__ float wave_sum_f(float v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, kWave);
    return v;
}

// The per-group exit test "wave-wide sum of v, plus rest, exceeds limit" (uniform result) in eight VALU
// instructions: four row rotations leave every lane of a 16-lane row with the row total, row_bcast:15 adds row 0's
// total into row 1 and row 2's into row 3, row_bcast:31 adds rows 0 + 1 into row 3 -- lane 63 then holds
// (r0 + r1) + (r2 + r3) -- and the comparison's bit 63 is the verdict.  Each step is ONE v_add_f32 with a DPP
// operand (the compiler leaves a v_mov_b32_dpp + v_add_f32 pair per step and four v_readlane); the two wait
// states a DPP read needs after the VALU write of its source are the s_nop 1 (inline asm is not hazard-checked).
// All 64 lanes must be active.
__device__ __forceinline__ float wave_sum_to_lane63(float v)
{
    float t;
    asm volatile("s_nop 1\n\t"
                 "v_add_f32_dpp %0, %1, %1 row_ror:8 row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 1\n\t"
                 "v_add_f32_dpp %0, %0, %0 row_ror:4 row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 1\n\t"
                 "v_add_f32_dpp %0, %0, %0 row_ror:2 row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 1\n\t"
                 "v_add_f32_dpp %0, %0, %0 row_ror:1 row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 1\n\t"
                 "v_add_f32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
                 "s_nop 1\n\t"
                 "v_add_f32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf"
                 : "=&v"(t) : "v"(v));
    return t;       // lane 63: (r0 + r1) + (r2 + r3); the other lanes hold partial sums
}

__device__ __forceinline__ bool wave_sum_exceeds(float v, float rest, float limit)
{
    return (__builtin_amdgcn_ballot_w64(wave_sum_to_lane63(v) + rest > limit) >> 63) != 0ull;
}

// Wave-wide sum, uniform result (all 64 lanes active), by the same six DPP adds.
__device__ __forceinline__ float wave_sum_all(float v)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, wave_sum_to_lane63(v)), 63));
}

// Inclusive suffix sums inside each 16-lane row (lane i <- sum of lanes i .. 15 of its row): four DPP adds.
// row_shl:n hands lane i the value of lane i + n; a lane whose source would lie outside the row keeps its value.
__device__ __forceinline__ float row_suffix_sum(float v)
{
    asm volatile("s_nop 1\n\t"
                 "v_add_f32_dpp %0, %0, %0 row_shl:1 row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 1\n\t"
                 "v_add_f32_dpp %0, %0, %0 row_shl:2 row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 1\n\t"
                 "v_add_f32_dpp %0, %0, %0 row_shl:4 row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 1\n\t"
                 "v_add_f32_dpp %0, %0, %0 row_shl:8 row_mask:0xf bank_mask:0xf"
                 : "+v"(v));
    return v;
}

#if MM_WITH_PRETIER
#include "filter_pretier.hpp"     // pretier_walk, pretier_canary (a fragment of this namespace)
#endif

// One pair through the fp32 form.  For CIEDE2000 `edge` reports a pair on the
// opposite-hue discontinuity (colour_diff_fast.hpp): the filter kernel logs it and
// edge_fix_kernel later replaces its fp32 term by the fp64 one.  cp, cq = chroma of
// p and q (read from the precomputed planes), lut = the hue table in LDS.
template <int TYPE, bool PREP = false>
__device__ __forceinline__ float pair_fast(const PxF& p, const PxF& q, float cp, float cq,
                                           const HueEntry* lut, bool* edge)
{
    if constexpr (TYPE == CIEDE2000) {
        return ciede2000_fast<PREP>(p, q, cp, cq, lut, edge);
    } else {
        *edge = false;
        return euclidean_fast(p, q);
    }
}

// The hue table, global -> LDS, by the whole workgroup.
__device__ __forceinline__ void load_hue_lut(HueEntry* lds, const HueEntry* __restrict__ global)
{
    for (int i = threadIdx.x; i < kHueLutEntries; i += blockDim.x) lds[i] = global[i];
    __syncthreads();
}

struct LibGroup { float4 q0, q1, q2, qc; };

constexpr float kChromaPrep = 0.02f;    // interleaved copies hold chroma / 50 (ciede2000_fast<PREP>)

// Angle in degrees wrapped into [-180, 180].
__device__ __forceinline__ float wrap180(float v) { return fmaf(-360.0f, rintf(v * (1.0f / 360.0f)), v); }

// Loads at (uniform base) + (32-bit byte offset): the form global_load takes with the
// base in SGPRs and one VGPR offset shared by all planes -- no 64-bit address arithmetic.
__device__ __forceinline__ float4 ld4(const float4* base, unsigned int byte_off)
{
    return *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(base) + byte_off);
}

// MM_LIB_NT (A/B builds): the library's groups as NON-TEMPORAL loads, on the idea that a visited library group is read once per (cell,
// candidate) while the cell's groups are read by every candidate of the item, so streaming the library past the L2 should leave the cells in
// it.  Measured (round 5, tools/experiments/archive/r5_nt.sh): config D +19 %, E's shard +22 %, I +16 %, C +13 %, G -2 % -- the cells of an XCD's
// sub-group DO share library lines in the L2 and the Infinity Cache, and the hint takes that away.  Off.
#ifndef MM_LIB_NT
#define MM_LIB_NT 0
#endif
__device__ __forceinline__ float4 ld4_lib(const float4* base, unsigned int byte_off)
{
    const float4* at = reinterpret_cast<const float4*>(reinterpret_cast<const char*>(base) + byte_off);
#if MM_LIB_NT
    typedef float f4v __attribute__((ext_vector_type(4)));
    const f4v v = __builtin_nontemporal_load(reinterpret_cast<const f4v*>(at));
    return float4{v.x, v.y, v.z, v.w};
#else
    return *at;
#endif
}

// NO_QC: the library's chroma plane is not fetched (filter_kernel<.., BIG>: a quarter of a visit's library bytes on cells whose launches
// press on the fabric); the caller recomputes it from a and b with the instructions that filled the plane (prep_chroma4), bit for bit.
template <int TYPE, bool NO_QC = false>
__device__ __forceinline__ LibGroup load_lib_group(const float4* b0, const float4* b1, const float4* b2,
                                                   const float4* bc, int q)
{
    const unsigned int off = (unsigned int)q * 16u;
    LibGroup g;
    g.q0 = ld4_lib(b0, off); g.q1 = ld4_lib(b1, off); g.q2 = ld4_lib(b2, off);
    if constexpr (TYPE == CIEDE2000 && !NO_QC) g.qc = ld4_lib(bc, off); else g.qc = float4{0.f, 0.f, 0.f, 0.f};
    return g;
}

// The interleaved copy's chroma plane of four pixels from their a and b: chroma_planes_kernel's chroma_of, then interleave_kernel's
// __fmul_rn by kChromaPrep (a product the compiler may not contract into a later sum) -- the stored values, bit for bit.
__device__ __forceinline__ float4 prep_chroma4(const float4& a, const float4& b)
{
    return float4{__fmul_rn(kChromaPrep, chroma_of(a.x, b.x)), __fmul_rn(kChromaPrep, chroma_of(a.y, b.y)),
                  __fmul_rn(kChromaPrep, chroma_of(a.z, b.z)), __fmul_rn(kChromaPrep, chroma_of(a.w, b.w))};
}

// Lower bound of  sum over a block's pixel pairs of dE00  from the seven statistics of the cell's block (c_*) and the
// library image's block (l_*): sum of channel 0, max |channel 0 - 50|, sums of channels 1 and 2, max chroma, hue arc
// centre (offset from 275 deg) and half-width (group_sums_kernel).  The derivation is in filter_kernel's comment.
// slack: what the fp32 rounding of the two sums can hide (1e-2 for the 64-term sums of values <= ~128 of a block).
__device__ __forceinline__ float stats_bound(float c_sl, float c_dev, float c_sa, float c_sb, float c_cmax, float c_hc, float c_hw,
                                             float l_sl, float l_dev, float l_sa, float l_sb, float l_cmax, float l_hc, float l_hw,
                                             float slack)
{
    // S_L of every pair of the block: |L_bar - 50| <= (max|L_cell - 50| + max|L_lib - 50|) / 2
    // over the block's active pixels, and S_L grows with it (:138-139)
    const float x = 0.5f * (c_dev + l_dev);
    const float x2 = x * x;
    const float s_l = fmaf(0.015f * x2, frsq(20.0f + x2), 1.0f);
    // S_C and R_T likewise from the largest chroma of the two blocks (see above)
    const float cbar = 0.5f * (c_cmax + l_cmax);
    // 1 + G of every pair is at least g_min: G falls as the pair's mean chroma grows (:50-53), and that is <= cbar
    const float g_min = 1.5f - 0.5f * sqrt_ratio7(cbar);
    const float gc = g_min * cbar;                                     // >= C_bar' of every pair
    const float s_c = fmaf(0.045f, gc, 1.0f);
    // |R_T| = sin(2 d_theta) R_C, d_theta = 30 deg * exp(-((h_bar' - 275) / 25)^2)  (:132-136,:145).  Each block's
    // hues lie on an arc (centre c as an offset from 275 deg, half-width w: statistics 5, 6); a pair's mean hue
    // is the midpoint of the shorter arc between its two hues (:107-122), i.e. within (w1 + w2) / 2 of
    // c1 + D / 2 with D = wrap(c2 - c1) -- or of the opposite point when some pair can be more than 180 deg
    // apart (|D| + w1 + w2 >= 180).  The reference does not wrap h_bar' - 275; the circular distance is
    // never larger, so this exponent is never more negative than the true one.
    const float c1 = c_hc, w1 = c_hw, c2 = l_hc, w2 = l_hw;
    const float delta = wrap180(c2 - c1);
    const float mid = fabsf(wrap180(fmaf(0.5f, delta, c1)));
    const float hw = 0.5f * (w1 + w2);
    const float d0 = fmaxf(0.0f, mid - hw), d1 = fmaxf(0.0f, (180.0f - mid) - hw);
    const float off = (fabsf(delta) + w1 + w2 >= 180.0f) ? fminf(d0, d1) : d0;   // least |h_bar' - 275| any pair can have
    const float e_max = fexp2(off * off * (-1.4426950408889634f / 625.0f));
    // v_sin_f32 takes revolutions: sin(60 deg * E) = sin(2 pi * E / 6); 1.001 + 1e-5 covers its error
    const float sin_2dt = fmaf(__builtin_amdgcn_sinf(e_max * (1.0f / 6.0f)), 1.001f, 1e-5f);
    const float half_rt = sqrt_ratio7(gc) * sin_2dt;                   // >= |R_T| / 2,  R_C = 2 sqrt_ratio7
    const float t_l = fmaxf(fabsf(c_sl - l_sl) - slack, 0.0f) * frcp(s_l);
    // dC'^2 + dH'^2 = ((1 + G) da)^2 + db^2 >= (g_min da)^2 + db^2
    const float da = fmaxf(fabsf(c_sa - l_sa) - slack, 0.0f) * g_min;
    const float db = fmaxf(fabsf(c_sb - l_sb) - slack, 0.0f);
    // tC^2 + tH^2 + R_T tC tH = (dC', dH') N (dC', dH')^T,  N = [[1/S_C^2, R_T/(2 S_C S_H)], [., 1/S_H^2]].
    // Its smallest eigenvalue falls as S_C, S_H or |R_T| grow (it is the minimum of y^T M y / (S_C^2 y1^2 +
    // S_H^2 y2^2) with M = [[1, R_T/2], [R_T/2, 1]] positive definite), so the block's worst constants give a
    // floor:  >= lambda (dC'^2 + dH'^2) >= lambda (da^2 + db^2);  S_H = 1 + 0.015 C' T <= 1 + 0.015 * 1.5726 g.
    const float inv_c = frcp(s_c), inv_h = frcp(fmaf(0.015f * 1.5726f, gc, 1.0f));
    const float na = inv_c * inv_c, nc = inv_h * inv_h, nb = half_rt * inv_c * inv_h;
    const float half_diff = 0.5f * (na - nc);
    // 0.999 and 0.9999: v_rsq / v_rcp / v_sqrt are good to ~1 ulp, and the difference of two close terms
    const float lambda = fmaxf(0.999f * (0.5f * (na + nc)) - fsqrt(fmaf(half_diff, half_diff, nb * nb)), 0.0f);
    return 0.9999f * fsqrt(fmaf(t_l, t_l, lambda * fmaf(da, da, db * db)));
}

// Work counters of a launch, at edge_count + kCounterBase (u32 indices; the u64 ones are 8-byte aligned):
//   +0,1  u64  256-pixel groups visited          +2,3  u64  groups a full visit of every candidate would take
//   +4,5  u64  candidates dropped by the block bound alone (no pixel touched)
//   +6    u32  unused                            words 32, 64, .. 256  u32  next item of XCD 0 .. 7 (persistent workgroups)
#ifndef MM_GRID_MULT
#define MM_GRID_MULT 1                // persistent workgroups launched per workgroup slot of the device (see persistent_grid)
#endif
#ifndef MM_SEED_GRID_CAP
#define MM_SEED_GRID_CAP 8192         // workgroups of seed_score_kernel at most (grid-stride over the (cell, seed) pairs)
#endif
#ifndef MM_LDS_GRID_CAP
#define MM_LDS_GRID_CAP 8192          // workgroups of filter_lds_kernel at most (they stride over the items)
#endif
#ifndef MM_STAT_BLOCKS
#define MM_STAT_BLOCKS (4 * kWave)   // blocks of a cell whose statistics filter_kernel keeps in LDS (experiments: 64 = cells up to 64 x 64 only)
#endif
#ifndef MM_BIG_RECOMPUTE_QC
#define MM_BIG_RECOMPUTE_QC 1
#endif
#ifndef MM_BIG_CELL_STAGE0
#define MM_BIG_CELL_STAGE0 1         // cells above 16 groups: stage 0 of the bound ahead of the candidate loop, bound-ordered visits from 17 groups on (0: round 3's form, for A/B builds)
#endif
#ifndef MM_LONG_SPLIT
#define MM_LONG_SPLIT 0.8f
#endif
constexpr float kLongSplit = MM_LONG_SPLIT;   // stage 2 of filter_kernel: candidates whose block bound is below this share of the limit go first
constexpr int kCounterBase = 2;
// (the item counters of the eight XCD shares sit on cache lines of their own: atomics on one line are served one after the other)
constexpr int kItemCounters = 32, kItemCounterStride = 32;
constexpr int kCounterWords = kFilterCounterWords;        // words cleared per launch (edge count, work counters, item counters)

// Whether candidate l of cell `cell` is one of the one-in-`every` (a power of two; 0: none) that audit_kernel checks
__device__ __forceinline__ bool audit_pick(unsigned int every, int cell, int l)
{
    return every != 0u && (((((unsigned int)cell * 0x9E3779B1u) ^ ((unsigned int)l * 0x85EBCA77u)) >> 7) & (every - 1u)) == 0u;
}

// XCD this wave runs on (0..7): hwreg(HW_REG_XCC_ID, 0, 4).  Used for L2 affinity only.
__device__ __forceinline__ int xcc_id() { return (int)(__builtin_amdgcn_s_getreg((3 << 11) | 20) & 7u); }

// Same work decomposition as score_kernel (kernels.hpp): one wave per
// (cell, candidate), lanes over float4 pixel groups, 4 waves = 4 candidates of
// one cell; cells fastest within a group so resident workgroups share library tiles.
//
// XCD-aware item order.  The cells of a launch are split into 8 contiguous shares, one per XCD; a
// persistent workgroup takes items of ITS XCD's share first (hwreg XCC_ID; a counter per XCD), in the
// order cells-fastest inside a sub-group of `cell_group` cells, so the workgroups sharing one 4 MB L2
// work on `cell_group` cells x a couple of candidate tiles at a time instead of all 8 L2s each seeing
// every cell and every tile of the launch.  A workgroup whose XCD has run dry steals from the next
// ones (the result does not depend on who scores what).
//
// CELL_LDS = true: the LDS-resident cell tile.  A larger workgroup (THREADS = 640: ten waves) stages the
// cell's three colour planes -- only the groups holding active pixels; channel 0 with NaN at inactive
// pixels -- in LDS once per item and every wave of it walks its candidates against that copy, so the
// global loads of the pixel loop are the library planes only.  The cell's chroma plane is not staged
// (it would be the 16 KB that keep a second workgroup off the CU): chroma is recomputed from a and b
// with the very instructions that filled the plane (chroma_of), bit for bit.  48 KB of cell + 16 KB
// hue table + statistics = 66 KB per workgroup: two workgroups = 20 waves per CU, 5 per SIMD.
//
// IL = true: interleaved images.  The pixel loop reads FOUR planes per side (channel 0, a, b, chroma); as four
// separate arrays they cost eight 64-bit base addresses in SGPRs per (cell, candidate), which is what pushed
// the kernel over the SGPR file (70 spilled, reloaded with ~55 v_readlane per candidate).  In the interleaved
// copy (interleave_kernel) an image is [group][plane][256 positions]: the four planes of a group are 1 KB
// apart, inside the 13-bit immediate offset of global_load, so ONE base per image and one VGPR offset
// address everything, and the groups a candidate visits before its early exit are one contiguous prefix.
// BIG: the instantiation for cells above 16 groups (80 x 80 .. 128 x 128): stage 0 of the bound ahead of the candidate loop and
// bound-ordered visits from 17 groups on.  A template parameter and not a run-time branch: the extra scalars it keeps alive
// cost the 64 x 64 instantiation three more spilled SGPRs and config D 0.3 % (measured, tools/experiments/archive/r4_second.sh).
// PRE: the packed-fp16 pre-tier ahead of the fp32 walk (filter_pretier.hpp; instantiated only with -DMM_WITH_PRETIER=1).
// How filter_kernel reads its arguments.  A persistent kernel with 27 arguments -- sixteen of them pointers, a `Variants` struct among them --
// keeps them all in SGPRs across its item loop, and the dominant instantiation then spilled 122 SGPRs into VGPR lanes and 4 VGPRs into
// scratch (20 bytes per lane): ~150 v_readlane reloads per item and ~27 per candidate.  The kernarg segment is memory the scalar unit reads
// through its own cache, so the arguments are read THERE at the point of use, through a pointer the compiler cannot see through (an empty
// asm: without it the loads are hoisted to the kernel's entry and the values held again): MM_KERNARG_RELOAD 2 (default) = every argument
// but check_every (the pixel loop's), 1 = the `Variants` struct only, 0 = round 5's form.  Dominant instantiation (-Rpass-analysis):
//   0: 122 spilled SGPRs, 4 spilled VGPRs, 20 B scratch, 95 VGPRs;  1: 63 / 0 / 20 / 95;  2: 14 / 0 / 0 / 91.
// Same box, two runs each (round 6): config D 43.19 -> 42.71 (1) / 42.66 ms per step (2), E's shard 29.22 -> 28.63 / 28.73, C 2.96 -> 2.93 /
// 2.94, G 14.08 -> 14.00 / 13.98; result digests equal.  FilterKernArgs mirrors the kernel's parameter list: the kernarg segment lays
// the explicit arguments out in order at their natural alignment, as a C struct -- whoever changes one changes the other (a mismatch
// garbles every argument: no test of the suite survives it).  The same treatment makes LATENCY-bound code slower -- a scalar load's ~50 cycles
// where a spill's v_readlane took 4: filter_lds_kernel<RGB> (stage 1: a candidate per lane, waiting on its own loads) 119 -> 16 spilled SGPRs
// and config B's kernel 0.102 -> 0.121 ms, F +3.5 %; resolve_front_kernel (one wave alone on its SIMD) 122 -> 83 and config PD's resolve
// 2.67 -> 3.15 ms: both left as they were.
#ifndef MM_KERNARG_RELOAD
#define MM_KERNARG_RELOAD 2
#endif
#if MM_KERNARG_RELOAD
struct FilterKernArgs {        // the kernel's explicit arguments as the kernarg segment lays them out (in order, naturally aligned)
    const float *cells, *cells_f0, *cell_chroma, *lib, *lib_chroma; float* scores;
    int n_cells, L, ppad, cand_per_block, cell_group;
    EdgeEntry* edges; unsigned int* edge_count; unsigned int edge_cap; const float* prune_above; int check_every;
    const HueEntry* hue_lut; Variants var; int xcd_shares; const int32_t* seeds; int n_seeds, sort_groups, item_chunk;
};
typedef const Variants __attribute__((address_space(4))) KernargVariants;
__device__ __forceinline__ KernargVariants* kernarg_variants()
{
    const char __attribute__((address_space(4)))* p = (const char __attribute__((address_space(4)))*)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(p));
    return (KernargVariants*)(p + offsetof(FilterKernArgs, var));
}
__device__ __forceinline__ const FilterKernArgs __attribute__((address_space(4)))* kernarg_all()
{
    const char __attribute__((address_space(4)))* p = (const char __attribute__((address_space(4)))*)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(p));
    return (const FilterKernArgs __attribute__((address_space(4)))*)p;
}
#define MM_VAR (*kernarg_variants())
#else
#define MM_VAR var
#endif
#if MM_KERNARG_RELOAD >= 2
#define KA(name) (kernarg_all()->name)
#else
#define KA(name) name
#endif

template <int TYPE, int THREADS, bool CELL_LDS, bool IL, bool BIG = false, bool PRE = false>
__global__ void __launch_bounds__(THREADS) __attribute__((amdgpu_waves_per_eu(MM_FILTER_WAVES, MM_FILTER_WAVES)))
filter_kernel(const float* __restrict__ cells, const float* __restrict__ cells_f0,
              const float* __restrict__ cell_chroma, const float* __restrict__ lib,
              const float* __restrict__ lib_chroma, float* __restrict__ scores,
              int n_cells, int L, int ppad, int cand_per_block, int cell_group,
              EdgeEntry* __restrict__ edges, unsigned int* __restrict__ edge_count,
              unsigned int edge_cap, const float* __restrict__ prune_above, int check_every,
              const HueEntry* __restrict__ hue_lut, Variants var, int xcd_shares,
              const int32_t* __restrict__ seeds, int n_seeds, int sort_groups, int item_chunk)
{
    // BIG cells (17 .. 64 groups: 80 .. 128 px) press on the fabric (config G 0.80 .. 0.95 of the 8 TB/s peak at a VALU rate of 0.40): their
    // library visits leave the stored chroma plane where it is and recompute it (two fmas, a v_sqrt and a multiply per pixel for a
    // quarter of the bytes).  -DMM_BIG_RECOMPUTE_QC=0: round 5's loads (A/B builds).
    constexpr bool kNoQc = BIG && IL && !CELL_LDS && TYPE == CIEDE2000 && MM_BIG_RECOMPUTE_QC != 0;
    __shared__ HueEntry lut[TYPE == CIEDE2000 ? kHueLutEntries : 1];
    __shared__ int seed_ids[256];           // the cell's KA(seeds): their KA(scores) are in the row already (launch_prune_limits)
    // stage 1 -> stage 2: the bound of every 256-position group of the tile's candidates (<= 64 per tile, <= 16 groups)
    __shared__ float cand_gb[TYPE == CIEDE2000 ? 64 : 1][TYPE == CIEDE2000 ? 16 : 1];
    // candidates of the tile that passed the bound in stage 1 (<= 64 per tile), in two classes: [0, 1] those whose bound is
    // below kLongSplit of the limit -- the likely contenders, which walk most of their groups -- and [2, 3] the others.
    // Stage 2 hands out the first class first: the waves of an item meet at a barrier, and a long candidate drawn last
    // keeps the other three waiting for it
    __shared__ unsigned int alive_mask[4];
    // stage 0 -> stage 1: the tile's candidates that passed the whole-image bound (and are no KA(seeds)), compacted
    __shared__ unsigned char surv_idx[TYPE == CIEDE2000 ? 64 : 1];
    __shared__ int n_surv;
    __shared__ float cell_ist[TYPE == CIEDE2000 ? 7 : 1];
    __shared__ int next_candidate;          // waves pull candidates: early exits make their cost uneven
    __shared__ unsigned int next_item;      // ... and workgroups pull (cell, tile) items: the hue table is loaded once
    __shared__ int item_share;              // the XCD share next_item belongs to (-1: none left)
    // block statistics of the cell (early-exit bound), up to 4 blocks per lane
    // (row 7: 1 where the block's mask pixels are all active in this cell, else 0 -- a clipped cell's bound counts only such blocks)
    __shared__ __attribute__((aligned(16))) float cell_st[TYPE == CIEDE2000 ? 8 : 1][TYPE == CIEDE2000 ? (CELL_LDS ? kWave : MM_STAT_BLOCKS) : 1];
    extern __shared__ float4 cell_tile[];   // CELL_LDS: [3][KA(ppad) / 4] -- channel 0 (NaN = inactive), a, b
    if constexpr (TYPE == CIEDE2000) load_hue_lut(lut, KA(hue_lut));
    const int n_tiles = (KA(L) + KA(cand_per_block) - 1) / KA(cand_per_block);
    const float* const lib_base = KA(lib);
    const float* const lib_chroma_base = KA(lib_chroma);
    unsigned int* const item_counter = KA(edge_count) + kItemCounters;
    const int my_share = KA(xcd_shares) > 1 ? xcc_id() % KA(xcd_shares) : 0;
    int first_live = 0;                     // thread 0: shares before my_share + first_live (cyclically) are exhausted
    // thread 0: items are taken `KA(item_chunk)` at a time -- one global atomic (a ~2 us round trip the whole workgroup sits
    // out) per chunk instead of per item; consecutive item numbers are neighbouring KA(cells) on the same candidate tile
    // (kept in LDS, not in registers of thread 0: the kernel sits exactly at its 96-VGPR budget)
    __shared__ unsigned int chunk_next;
    __shared__ int chunk_left, chunk_share;
    if (threadIdx.x == 0) chunk_left = 0;
#ifdef MM_FILTER_STAMPS
    // Instrumented build (tools/experiments/archive/r3_stamps.sh): where a wave's cycles go inside an item, summed over the launch
    // into words 24..31 of the counters (units of 1024 cycles): 0 waiting for the workgroup's slowest wave, 1 item fetch
    // (the global atomic), 2 statistics / KA(seeds) into LDS, 3 stage 0, 4 stage 1, 5 stage 2; 6 the longest workgroup lifetime, 7 the sum of the lifetimes / 16
    __shared__ unsigned int stamp_acc[8];
    if (threadIdx.x < 8) stamp_acc[threadIdx.x] = 0u;
    unsigned long long stamp_t = __builtin_readcyclecounter();
    const unsigned long long stamp_begin = stamp_t;
    unsigned int stamp_items = 0;
#define MM_STAMP(k) do { const unsigned long long now_ = __builtin_readcyclecounter(); \
        if ((threadIdx.x & 63) == 0) atomicAdd(&stamp_acc[k], (unsigned int)((now_ - stamp_t) >> 4)); stamp_t = now_; } while (0)
#else
#define MM_STAMP(k) do { } while (0)
#endif
    unsigned int n_visited = 0, n_groups_all = 0, n_bound_dropped = 0;     // work counters, per lane (a wave's share of a launch fits 32 bits)
    // Persistent workgroups: as many as the device holds at once, each taking the next item of the
    // launch order from a global counter.  A 33 KB table per 32-candidate tile was
    // three times the HBM traffic of the launch in L2 -> LDS copies, and its latency idled the waves
    // at every workgroup start.
    for (;;) {
    __syncthreads();                        // the previous item is done with next_candidate / cell_st
    MM_STAMP(0);
    if (threadIdx.x == 0) {
        int share = -1;
        unsigned int item = 0;
        if (chunk_left > 0) {
            share = chunk_share; item = chunk_next; chunk_next = item + 1u; chunk_left = chunk_left - 1;
        } else {
            for (int k = first_live; k < KA(xcd_shares); ++k) {
                const int sh = (my_share + k) % KA(xcd_shares);
                const long long lo = (long long)KA(n_cells) * sh / KA(xcd_shares), hi = (long long)KA(n_cells) * (sh + 1) / KA(xcd_shares);
                const long long total = (hi - lo) * n_tiles;
                const unsigned int it = atomicAdd(item_counter + sh * kItemCounterStride, (unsigned int)KA(item_chunk));
                if ((long long)it < total) {
                    share = sh; item = it;
                    chunk_share = sh; chunk_next = it + 1u;
                    chunk_left = (int)min((long long)KA(item_chunk), total - (long long)it) - 1;
                    break;
                }
                first_live = k + 1;             // that share is exhausted for good
            }
        }
        next_item = item;
        item_share = share;
        next_candidate = THREADS / kWave;
    }
    __syncthreads();
    MM_STAMP(1);
    // read back as scalars: everything derived from them (cell, tile, plane addresses) stays in SGPRs
    const int share = __builtin_amdgcn_readfirstlane(item_share);
    if (share < 0) break;
    const long long bid = (long long)(unsigned int)__builtin_amdgcn_readfirstlane((int)next_item);
    const int share_lo = (int)((long long)KA(n_cells) * share / KA(xcd_shares));
    const int share_cells = (int)((long long)KA(n_cells) * (share + 1) / KA(xcd_shares)) - share_lo;
    const long long per_group = (long long)KA(cell_group) * n_tiles;
    const int group = (int)(bid / per_group);
    const long long rem = bid - (long long)group * per_group;
    const int cells_here = min(KA(cell_group), share_cells - group * KA(cell_group));
    const int tile = (int)(rem / cells_here);
    const int cell_at = share_lo + group * KA(cell_group) + (int)(rem - (long long)tile * cells_here);
    if (tile >= n_tiles || cell_at >= KA(n_cells)) continue;
    // (long items first: launch_filter's order of the share's KA(cells); a scalar load, the cell number stays in an SGPR)
    const int cell = MM_VAR.cell_order ? __builtin_amdgcn_readfirstlane(MM_VAR.cell_order[cell_at]) : cell_at;

    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int v = MM_VAR.cell_var[cell];
    const int groups = MM_VAR.groups[v];                    // only the variant's active pixels
    lib = lib_base + (size_t)v * (IL ? MM_VAR.chroma_stride * 4 : MM_VAR.lib_stride);
    lib_chroma = IL ? nullptr : lib_chroma_base + (size_t)v * MM_VAR.chroma_stride;
    // channel 0 from the plane that holds NaN at inactive pixels: their terms come out as
    // max(NaN, 0) = 0 (colour_diff_fast.hpp), so there is no mask to load or test here
    // IL: KA(cells_f0) is the interleaved copy [cell][group][4][256]; planes of a group 64 float4 apart
    const float4* c0 = reinterpret_cast<const float4*>(KA(cells_f0) + (size_t)cell * (IL ? 4 : 1) * KA(ppad));
    const float4* c1 = IL ? c0 + kWave : reinterpret_cast<const float4*>(KA(cells) + (size_t)cell * 3 * KA(ppad)) + KA(ppad) / 4;
    const float4* c2 = IL ? c0 + 2 * kWave : c1 + KA(ppad) / 4;
    const float4* cc = IL ? c0 + 3 * kWave : reinterpret_cast<const float4*>(KA(cell_chroma) + (size_t)cell * KA(ppad));
    const int pq = KA(ppad) / 4;
    for (int t = threadIdx.x; t < KA(n_seeds); t += THREADS) seed_ids[t] = KA(seeds)[(size_t)cell * KA(n_seeds) + t];
    if constexpr (CELL_LDS) {                             // stage the cell (active groups only); the barrier below covers it
        const int nq = groups * kWave;
        for (int q = threadIdx.x; q < nq; q += THREADS) {
            const int src = IL ? (q & (kWave - 1)) + 4 * kWave * (q >> 6) : q;      // interleaved: group q / 64, position q % 64
            cell_tile[q] = c0[src];
            cell_tile[pq + q] = c1[src];
            cell_tile[2 * pq + q] = c2[src];
        }
    }

    // Early exit, the device form of the reference's `variant < bestVariant` loop guards
    // (CPUPhotomosaicGenerator.cpp:149,153): every term is >= 0, so once the running
    // sum of a candidate exceeds `limit` -- an upper bound on the score a contender of
    // this cell can have (prune_limit_kernel) -- the candidate cannot be one and gets +inf.
    // The test is wave-uniform: one wave owns the whole candidate.
    //
    // CIEDE2000 exits earlier still, through a lower bound on the pixels NOT yet visited:
    //   dE00^2 = tL^2 + tC^2 + tH^2 + R_T tC tH >= tL^2 + (1 - |R_T|/2) (tC^2 + tH^2)        (:153-157)
    //   tC^2 + tH^2 >= (dC'^2 + dH'^2) / S_C^2            (S_H = 1 + 0.015 C' T <= S_C = 1 + 0.045 C', T < 1.93)
    //   dC'^2 + dH'^2 = |(a1', b1) - (a2', b2)|^2 = ((1 + G) da)^2 + db^2 >= (g_min da)^2 + db^2   (a' = (1 + G) a, G >= G(largest mean chroma))
    // so per pair  dE00 >= sqrt( (dL / S_L)^2 + rho (da^2 + db^2) / S_C^2 ),  rho = 1 - |R_T| / 2 (sharpened below to the
    // smallest eigenvalue of the chroma / hue form, which also knows that S_H < S_C), and with the
    // worst constants of a 256-pixel group g of the plane order in place of the per-pair ones:
    //   S_L <= S_L((max_g |L_cell - 50| + max_g |L_lib - 50|) / 2)         (the pair's mean lightness lies within; :138-139)
    //   C_bar' <= g(C_bar) = (1.5 - 0.5 sqrt(C_bar^7 / (C_bar^7 + 25^7))) C_bar, increasing, C_bar <= (max_g C_cell + max_g C_lib) / 2
    //   S_C <= 1 + 0.045 g(.),   |R_T| <= sin(60 deg * E) * 2 sqrt(g^7 / (g^7 + 25^7)),  E = the largest exp(-((h_bar' - 275)/25)^2)
    //   the two blocks' hue ranges allow (below; :132-136,:145)
    // and, by the triangle inequality of the Euclidean norm over the group's pixels,
    //   sum_{p in g} dE00 >= sqrt( (sum|dL|)^2 / S_L^2 + rho ((sum|da|)^2 + (sum|db|)^2) / S_C^2 ),  sum_p |dX| >= |sum_p X_cell - sum_p X_lib|,
    // which needs SEVEN precomputed numbers per image and block (group_sums_kernel) -- no pass
    // over the pixels.  Lane g holds group g's bound; a suffix scan turns it into "bound
    // of groups >= j".  The candidate is dropped once
    //   (trusted partial sum of groups < j) + (bound of groups >= j) > limit.
    // The library's block statistics are over the mask's pixels, so a block counts only where the cell's active pixels
    // ARE the mask's (statistic 7 of the cell, gather_prepare_kernel): all blocks of an unclipped cell, the blocks wholly
    // inside the bound of a clipped one -- its other blocks contribute 0, which every sum of non-negative terms exceeds.
    // Terms of pairs flagged for the fp64 patch (hue discontinuity) are summed apart and
    // never enter the exit test: only terms with a proven error bound may prune.
    const float limit = KA(prune_above) ? KA(prune_above)[cell] : __builtin_huge_valf();
    bool use_lb = false;
    if constexpr (TYPE == CIEDE2000) {
        use_lb = KA(prune_above) != nullptr && MM_VAR.cell_full[cell] != 0 && groups <= (CELL_LDS ? kWave / 4 : kWave);     // workgroup-uniform
        if (use_lb) {
            // the cell's seven numbers per block (group_sums_kernel) wait in LDS: read per candidate, no registers held
            const bool clipped = MM_VAR.cell_full[cell] == 2;
            for (int t = threadIdx.x; t < 4 * groups; t += THREADS) {
                for (int k = 0; k < 7; ++k)
                    cell_st[k][t] = MM_VAR.cell_stats[(size_t)k * MM_VAR.cell_stats_stride + (size_t)cell * MM_VAR.n_blocks + t];
                // (statistic 7, the factor of every block's bound, carries the diagnostics' scale of the bound: 1)
                cell_st[7][t] = MM_VAR.bound_scale * (clipped ? MM_VAR.cell_stats[(size_t)7 * MM_VAR.cell_stats_stride + (size_t)cell * MM_VAR.n_blocks + t] : 1.0f);
            }
        }
    }
    // blocks per lane in the bound's prelude: 1, 2 or 4 (a power of two, so groups start on lane boundaries)
    const int lane_shift = groups <= 16 ? 0 : (groups <= 32 ? 1 : 2), per_lane = 1 << lane_shift;
    const int l_base = tile * KA(cand_per_block);
    const int n_here = min(KA(cand_per_block), KA(L) - l_base);
    // Two stages.  The bound of a candidate takes one stats_bound per 64-position block (~75 instructions, ten of them
    // transcendental) and half of the candidates end there, so it is computed where all 64 lanes work on it.  Stage 1:
    // sixteen lanes hold a candidate, lane g its group g = four blocks, so one pass of the wave judges FOUR candidates;
    // the waves share the tile's candidates four at a time, leave every candidate's sixteen group bounds in LDS and a
    // mask of the survivors.  Stage 2: the waves pull survivors from the mask; a survivor's prelude is then a read of its
    // group bounds, the visiting order and a scan -- ~60 instructions instead of ~300 -- before the pixel loop.
    bool two_level = false;
    if constexpr (TYPE == CIEDE2000) two_level = use_lb && lane_shift == 0 && groups > 0 && KA(cand_per_block) <= 64 && (MM_VAR.two_stage & 1) != 0;
    if (two_level && threadIdx.x < 4) alive_mask[threadIdx.x] = 0u;
    // Stage 0: ONE bound per candidate from the statistics of the two WHOLE images (the triangle inequality holds for any
    // partition of the pixels, the one-block partition included).  Weaker than the block bound, but 1 / 64 of its cost, and it
    // drops a third of config D's candidates (more on smoother content) before stage 1 spends 64 block evaluations on them.
    // Unclipped KA(cells) only (the image statistics are over the whole mask).
    bool stage0 = false;
    if constexpr (TYPE == CIEDE2000) {
        // (KA(cells) above 16 groups have no stage 1 -- a lane of the prelude holds two or four blocks there -- but stage 0 is the same
        // one pass of wave 0 whatever the cell size: the survivors' compacted list then feeds the candidate loop directly)
        stage0 = use_lb && groups > 0 && KA(cand_per_block) <= 64 && (MM_VAR.two_stage & 2) != 0 && MM_VAR.cell_full[cell] == 1 &&
                 (two_level || (BIG && MM_BIG_CELL_STAGE0 != 0 && lane_shift > 0));
        // (the image-level statistics sit in slot KA(ppad) / 64 of each image's block statistics: no arguments of their own)
        if (stage0 && threadIdx.x < 7)
            cell_ist[threadIdx.x] = MM_VAR.cell_stats[(size_t)threadIdx.x * MM_VAR.cell_stats_stride + (size_t)cell * MM_VAR.n_blocks + KA(ppad) / 64];
    }
    if (use_lb || CELL_LDS || KA(n_seeds) > 0) __syncthreads();   // workgroup-uniform
    MM_STAMP(2);
    if constexpr (TYPE == CIEDE2000) if (stage0) {
        if (wave == 0) {
            const int l = l_base + lane;
            const bool valid = lane < n_here;
            bool seed = false;
            if (valid)
                for (int t = 0; t < KA(n_seeds); ++t) seed = seed || seed_ids[t] == l;
            bool keep = valid && !seed;
            if (keep) {
                const float* is = MM_VAR.lib_stats + ((size_t)v * KA(L) + l) * MM_VAR.n_blocks + KA(ppad) / 64;
                const size_t ss = MM_VAR.lib_stats_stride;
                // slack 1: the fp32 rounding of a 4096-term sum of values <= 128 kept in fp64 until its one rounding (0.03),
                // plus what the 64 block sums behind it carry (64 x 1e-2)
                const float b0 = stats_bound(cell_ist[0], cell_ist[1], cell_ist[2], cell_ist[3], cell_ist[4], cell_ist[5], cell_ist[6],
                                             is[0], is[ss], is[2 * ss], is[3 * ss], is[4 * ss], is[5 * ss], is[6 * ss], 1.0f);
                n_groups_all += (unsigned int)groups;
                if (b0 * MM_VAR.bound_scale > limit) {
                    KA(scores)[(size_t)cell * KA(L) + l] = __builtin_huge_valf();
                    n_bound_dropped += 1;
                    keep = false;
                }
            }
            const unsigned long long kept = __builtin_amdgcn_ballot_w64(keep);
            if (keep) surv_idx[__builtin_popcountll(kept & ((1ull << lane) - 1ull))] = (unsigned char)lane;
            if (lane == 0) n_surv = __builtin_popcountll(kept);
        }
        __syncthreads();
    }
    MM_STAMP(3);
    if constexpr (TYPE == CIEDE2000) if (two_level) {         // (the statistics arrays exist for CIEDE2000 only)
        const int n_stage1 = stage0 ? n_surv : n_here;
        // lanes (q, g): candidate q of the pass, group g < groups -- 64 / groups candidates per pass (four at 16 groups,
        // seven at the 9 groups of a 48 x 48 cell), packed so that no lane idles on a group the cell does not have
        const int per_pass = kWave / groups;
        const int q = lane / groups, g = lane - q * groups;
        for (int first = per_pass * wave; first < n_stage1; first += per_pass * (THREADS / kWave)) {
            const bool valid = q < per_pass && first + q < n_stage1;
            const int idx = stage0 ? (valid ? (int)surv_idx[first + q] : 0) : first + q;      // stage 0 left a compacted list
            const int l = l_base + idx;
            float b = 0.0f;
            // scored already as a seed: not a survivor.  The candidate's lanes share the seed list out; one ballot gathers the hits
            // (stage 0 has taken the KA(seeds) out already)
            bool hit = false;
            if (valid && !stage0)
                for (int t = g; t < KA(n_seeds); t += groups) hit = hit || seed_ids[t] == l;
            const unsigned long long hits = __builtin_amdgcn_ballot_w64(hit);
            const bool seed = ((hits >> (q * groups)) & ((1ull << groups) - 1ull)) != 0ull;     // (q * groups < 64 where it matters: valid lanes)
            if (valid) {
                // the group's four blocks are four consecutive floats of each statistic
                const float* st = MM_VAR.lib_stats + ((size_t)v * KA(L) + l) * MM_VAR.n_blocks + 4 * g;
                const size_t bs = MM_VAR.lib_stats_stride;
                float4 ls[7];
                for (int k = 0; k < 7; ++k) ls[k] = *reinterpret_cast<const float4*>(st + k * bs);
                auto block = [&](int i, float l0, float l1, float l2, float l3, float l4, float l5, float l6) {
                    const int blk = 4 * g + i;
                    b += cell_st[7][blk] * stats_bound(cell_st[0][blk], cell_st[1][blk], cell_st[2][blk], cell_st[3][blk], cell_st[4][blk], cell_st[5][blk], cell_st[6][blk],
                                                       l0, l1, l2, l3, l4, l5, l6, 1e-2f);
                    __builtin_amdgcn_sched_barrier(0);  // one block at a time: four interleaved evaluations do not fit the registers
                };
                block(0, ls[0].x, ls[1].x, ls[2].x, ls[3].x, ls[4].x, ls[5].x, ls[6].x);
                block(1, ls[0].y, ls[1].y, ls[2].y, ls[3].y, ls[4].y, ls[5].y, ls[6].y);
                block(2, ls[0].z, ls[1].z, ls[2].z, ls[3].z, ls[4].z, ls[5].z, ls[6].z);
                block(3, ls[0].w, ls[1].w, ls[2].w, ls[3].w, ls[4].w, ls[5].w, ls[6].w);
                cand_gb[idx][g] = b;                        // (stage 2 reads the first `groups` entries only)
            }
            // sum over the candidate's lanes: a suffix scan that stops at the segment's end; its first lane gets the total
            if (groups == 16) {
                b = row_suffix_sum(b);                      // a candidate is a 16-lane row: four DPP adds
            } else {
                for (int off = 1; off < 16; off <<= 1) {
                    const float up = __shfl_down(b, off, kWave);
                    if (g + off < groups) b += up;
                }
            }
            if (g == 0 && valid && !seed) {
                if (b > limit) {
                    KA(scores)[(size_t)cell * KA(L) + l] = __builtin_huge_valf();
                    n_bound_dropped += 1;       // (per lane; summed over the wave's lanes at the end)
                } else {
                    atomicOr(&alive_mask[(b < kLongSplit * limit ? 0 : 2) + (idx >> 5)], 1u << (idx & 31));
                }
                if (!stage0) n_groups_all += (unsigned int)groups;     // (stage 0 counted its candidates itself)
            }
        }
        if (threadIdx.x == 0) next_candidate = 0;           // stage 2 counts survivors from 0
        __syncthreads();
    }
    MM_STAMP(4);
    // (the masks are read from LDS again per candidate: four more SGPRs held across the pixel loop cost more in spills)
    const int n_long = two_level ? __builtin_popcount(alive_mask[0]) + __builtin_popcount(alive_mask[1]) : 0;
    const int n_todo = two_level ? n_long + __builtin_popcount(alive_mask[2]) + __builtin_popcount(alive_mask[3]) : (BIG && stage0 ? n_surv : n_here);
    for (int turn = two_level ? 0 : wave; ;) {
        if (two_level) {                                    // the next survivor not yet taken by any wave
            int nxt = 0;
            if (lane == 0) nxt = atomicAdd(&next_candidate, 1);
            turn = __builtin_amdgcn_readfirstlane(nxt);
        }
        if (turn >= n_todo) break;
        int idx = turn;
        if (two_level) {                                    // the turn-th set bit of the masks, the long class first
            const int cls = turn < n_long ? 0 : 2;
            const unsigned long long survivors = ((unsigned long long)alive_mask[cls + 1] << 32) | alive_mask[cls];
            const int nth = turn < n_long ? turn : turn - n_long;
            const bool mine = ((survivors >> lane) & 1ull) != 0 && __builtin_popcountll(survivors & ((1ull << lane) - 1ull)) == nth;
            idx = __builtin_ctzll(__builtin_amdgcn_ballot_w64(mine));
        } else if (BIG && stage0) {
            idx = __builtin_amdgcn_readfirstlane((int)surv_idx[turn]);     // stage 0's compacted list (KA(seeds) are not on it)
        }
        const int l = l_base + idx;
        if (!two_level) {   // the next candidate of this tile not yet taken by any wave
            int nxt = 0;
            if (lane == 0) nxt = atomicAdd(&next_candidate, 1);
            turn = __builtin_amdgcn_readfirstlane(nxt);      // scalar: library addresses stay in SGPRs
            if (KA(n_seeds) > 0 && !(BIG && stage0)) {
                // scored already as a seed (seed_score_kernel left its score in the row): nothing to do
                bool hit = false;
                for (int t = lane; t < KA(n_seeds); t += kWave) hit = hit || seed_ids[t] == l;
                if (__builtin_amdgcn_ballot_w64(hit) != 0) continue;
            }
        }
        const float4* b0 = reinterpret_cast<const float4*>(lib + (size_t)l * (IL ? 4 : 3) * KA(ppad));
        const float4* b1 = IL ? b0 + kWave : b0 + KA(ppad) / 4;
        const float4* b2 = IL ? b0 + 2 * kWave : b1 + KA(ppad) / 4;
        const float4* bc = IL ? b0 + 3 * kWave : reinterpret_cast<const float4*>(lib_chroma + (size_t)l * KA(ppad));
        bool pruned = false;
        float rest = 0.0f;                 // lane k: bound of the groups not yet visited at step k (see rest_at)
        int order = lane;                  // lane k: the group visited k-th
        const bool sorted = TYPE == CIEDE2000 && use_lb && lane_shift == 0 && KA(sort_groups) != 0;
        // KA(cells) of 33 .. 64 groups (up to 128 x 128): a lane of the prelude holds exactly one group (four blocks), so the
        // same descending-bound visiting order comes from ranking the 64 lanes against each other
        // ... and KA(cells) of 17 .. 32 groups (72 .. 90 px), whose lanes hold HALF a group each: the halves are added up into lanes
        // 0 .. 31 first (two ds_bpermute), then ranked the same way
        const bool sorted64 = TYPE == CIEDE2000 && use_lb && lane_shift >= (BIG && MM_BIG_CELL_STAGE0 != 0 ? 1 : 2) && KA(sort_groups) != 0 && !two_level;
        const bool by_group = sorted || two_level || sorted64;      // lane k of `rest`: visiting step k (else: block k)
        if constexpr (TYPE == CIEDE2000) {
            if (use_lb) {
                // Lane k takes blocks k*m .. k*m+m-1 (m = blocks per lane: 1 up to 64 x 64 KA(cells), so a lane
                // is a quarter of a group; 4 at 128 x 128: a lane is a group).  After stage 1 the group bounds wait in LDS.
                const int n_blk = 4 * groups;
                float d = 0.0f;
                if (two_level) {
                    if (!sorted) d = lane < groups ? cand_gb[idx][lane] : 0.0f;   // lane k: bound of group k
                } else {
                    for (int i = 0; i < per_lane; ++i) {
                        const int blk = lane * per_lane + i;
                        if (blk >= n_blk) break;
                        const float* st = MM_VAR.lib_stats + ((size_t)v * KA(L) + l) * MM_VAR.n_blocks + blk;
                        const size_t bs = MM_VAR.lib_stats_stride;
                        d += cell_st[7][blk] * stats_bound(cell_st[0][blk], cell_st[1][blk], cell_st[2][blk], cell_st[3][blk], cell_st[4][blk], cell_st[5][blk], cell_st[6][blk],
                                                           st[0], st[bs], st[2 * bs], st[3 * bs], st[4 * bs], st[5 * bs], st[6 * bs], 1e-2f);
                    }
                }
                if (sorted) {
                    // Visit the groups in DESCENDING order of their bounds.  The order is free -- the test is "sum of the
                    // visited groups + bound of the unvisited ones > limit" whichever groups were visited -- and a group's
                    // true sum exceeds its bound roughly in proportion to the bound, so the largest bounds are replaced by
                    // true sums first and the candidate is dropped ~12 % of the visits earlier (simulated on config D).
                    // One block per lane here (groups <= 16): quad j of the wave = group j.
                    auto quad_swap = [](float x, int ctrl_tag) {
                        const int r = ctrl_tag == 1 ? __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, x), 0xB1, 0xf, 0xf, false)    // quad_perm [1,0,3,2]
                                                    : __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, x), 0x4E, 0xf, 0xf, false);   // quad_perm [2,3,0,1]
                        return __builtin_bit_cast(float, r);
                    };
                    float g;                                                    // every lane of quad j: bound of group j (0 past the last group)
                    if (two_level) {
                        g = (lane >> 2) < groups ? cand_gb[idx][lane >> 2] : 0.0f;
                    } else {
                        g = d + quad_swap(d, 1);
                        g += quad_swap(g, 2);
                    }
                    const int j = lane >> 2, t = lane & 3;
                    // unique integer keys: the bound's bits (non-negative floats order like integers) with the group number
                    // in the low 4 bits, lower group first among equals
                    const int key = (__builtin_bit_cast(int, g) & ~15) | (15 - j);
                    // lane (j, t) counts the groups 4 t .. 4 t + 3 that come before group j; the quad adds its four counts up
                    const int pull = t * 64;                                    // byte address of lane 16 t = quad 4 t
                    int before = 0;
                    before += __builtin_amdgcn_ds_bpermute(pull, key) > key;
                    before += __builtin_amdgcn_ds_bpermute(pull + 16, key) > key;
                    before += __builtin_amdgcn_ds_bpermute(pull + 32, key) > key;
                    before += __builtin_amdgcn_ds_bpermute(pull + 48, key) > key;
                    before += __builtin_amdgcn_mov_dpp(before, 0xB1, 0xf, 0xf, false);
                    before += __builtin_amdgcn_mov_dpp(before, 0x4E, 0xf, 0xf, false);     // rank of group j (groups past the last have the smallest keys)
                    // Only the `KA(sort_groups)` largest bounds are visited out of order; the other groups follow in plane order
                    // (position = groups of lower number that are not among the largest), which keeps most of a candidate's
                    // reads a contiguous run -- the L2 hit rate falls when every candidate walks its own permutation.
                    const unsigned long long tail_lanes = __builtin_amdgcn_ballot_w64(t == 0 && before >= KA(sort_groups));
                    const int below = __builtin_popcountll(tail_lanes & ((1ull << lane) - 1ull));
                    const int pos = before < KA(sort_groups) ? before : KA(sort_groups) + below;
                    // scatter (group number, bound) to the lane of their place; every lane pushes somewhere unique
                    const int dest = t != 0 ? 16 + 3 * j + t - 1 : pos;
                    order = __builtin_amdgcn_ds_permute(dest * 4, j);
                    d = __builtin_bit_cast(float, __builtin_amdgcn_ds_permute(dest * 4, __builtin_bit_cast(int, g)));
                    if (lane >= 16) d = 0.0f;                                   // lanes 0..15: bounds in visiting order
                }
                if (sorted64) {
                    if (lane_shift == 1) {      // lane j < 32 <- group j = the blocks of lanes 2 j and 2 j + 1
                        const float lo = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute((2 * lane) * 4, __builtin_bit_cast(int, d)));
                        const float hi = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute((2 * lane + 1) * 4, __builtin_bit_cast(int, d)));
                        d = lane < kWave / 2 ? lo + hi : 0.0f;
                    }
                    // unique keys: the bound's bits (non-negative floats order like integers) with the lane number in the low
                    // 6 bits, lower group first among equals; lanes past the last group hold 0 and keep their places at the end
                    const int key = (__builtin_bit_cast(int, d) & ~63) | (63 - lane);
                    int before = 0;
#pragma nounroll
                    for (int j = 0; j < kWave; ++j) before += __builtin_amdgcn_readlane(key, j) > key;
                    // as above: only the `KA(sort_groups)` largest bounds are visited out of order, the rest in plane order
                    const unsigned long long tail_lanes = __builtin_amdgcn_ballot_w64(before >= KA(sort_groups));
                    const int below = __builtin_popcountll(tail_lanes & ((1ull << lane) - 1ull));
                    const int pos = before < KA(sort_groups) ? before : KA(sort_groups) + below;
                    order = __builtin_amdgcn_ds_permute(pos * 4, lane);
                    d = __builtin_bit_cast(float, __builtin_amdgcn_ds_permute(pos * 4, __builtin_bit_cast(int, d)));
                }
                // inclusive suffix scan over lanes (lane j <- sum of lanes >= j); group bounds fill lanes 0..15 only
                if (sorted || two_level) {
                    d = row_suffix_sum(d);
                } else {
                    for (int off = 1; off < kWave; off <<= 1) {
                        const float up = __shfl_down(d, off, kWave);
                        if (lane + off < kWave) d += up;
                    }
                }
                rest = d;
                pruned = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, rest), 0)) > limit;
            }
        }
        int visited = 0;
        float pre_sum = -1.0f;         // PRE: lane-partial fp16-form sum of a candidate that survived the pre-tier (else < 0)
#if MM_WITH_PRETIER
        if constexpr (PRE) {
            if (!pruned && groups > 0 && KA(prune_above) != nullptr)
                pruned = pretier_walk(reinterpret_cast<const char*>(KA(cells)) + (size_t)cell * 8 * KA(ppad),
                                      reinterpret_cast<const char*>(lib_chroma_base) + ((size_t)v * KA(L) + l) * 8 * (size_t)KA(ppad),
                                      lane, groups, check_every, use_lb, rest, by_group, lane_shift, sorted || sorted64, order,
                                      limit, MM_VAR.pre_abs, MM_VAR.pre_rel, lut, visited, pre_sum);
        }
#endif
        float acc = 0.0f;          // terms with a proven error bound
        float acc_edge = 0.0f;     // terms awaiting the fp64 patch
        if (!pruned && groups > 0) {
            // One 256-pixel group of the candidate against the cell; true = candidate dropped.
            auto visit = [&](const LibGroup& cur, int j, int step) -> bool {
                if (step != 0 && step % check_every == 0) {
                    // bound of what is still unvisited: sorted, lane `step` of the scan over the visiting order; else group
                    // `step` starts at block 4 step = the first block of lane (4 step) >> lane_shift
                    const float rest_j = use_lb ? __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, rest), by_group ? step : (4 * step) >> lane_shift)) : 0.0f;
                    if (wave_sum_exceeds(acc, rest_j, limit)) return true;
                }
                const int q = lane + kWave * j;
                // byte offset of this lane's float4 in plane 0 of group j (interleaved: a group is 4 KB, planes 1 KB apart)
                const unsigned int off = IL ? (unsigned int)lane * 16u + (unsigned int)j * 4096u : (unsigned int)q * 16u;
                visited += 1;
                float4 p0, p1, p2, pc{0.f, 0.f, 0.f, 0.f};
                if constexpr (CELL_LDS) {
                    p0 = cell_tile[q]; p1 = cell_tile[pq + q]; p2 = cell_tile[2 * pq + q];
                    if constexpr (TYPE == CIEDE2000) {
                        pc = float4{chroma_of(p1.x, p2.x), chroma_of(p1.y, p2.y), chroma_of(p1.z, p2.z), chroma_of(p1.w, p2.w)};
                        if constexpr (IL)    // as interleave_kernel stores it (__fmul_rn: a product the compiler may not contract into a later sum)
                            pc = float4{__fmul_rn(kChromaPrep, pc.x), __fmul_rn(kChromaPrep, pc.y), __fmul_rn(kChromaPrep, pc.z), __fmul_rn(kChromaPrep, pc.w)};
                    }
                } else {
                    p0 = ld4(c0, off); p1 = ld4(c1, off); p2 = ld4(c2, off);
                    if constexpr (TYPE == CIEDE2000) pc = ld4(cc, off);
                }
                const float4 q0 = cur.q0, q1 = cur.q1, q2 = cur.q2, qc = kNoQc ? prep_chroma4(cur.q1, cur.q2) : cur.qc;
                bool e0, e1, e2, e3;
                const float v0 = pair_fast<TYPE, IL>(PxF{p0.x, p1.x, p2.x}, PxF{q0.x, q1.x, q2.x}, pc.x, qc.x, lut, &e0);
                const float v1 = pair_fast<TYPE, IL>(PxF{p0.y, p1.y, p2.y}, PxF{q0.y, q1.y, q2.y}, pc.y, qc.y, lut, &e1);
                const float v2 = pair_fast<TYPE, IL>(PxF{p0.z, p1.z, p2.z}, PxF{q0.z, q1.z, q2.z}, pc.z, qc.z, lut, &e2);
                const float v3 = pair_fast<TYPE, IL>(PxF{p0.w, p1.w, p2.w}, PxF{q0.w, q1.w, q2.w}, pc.w, qc.w, lut, &e3);
                if constexpr (TYPE == CIEDE2000) {
                    // discontinuity pairs are rare (~1e-5): one wave-wide test, then the slow branch
                    // not NaN = active; the interleaved cell carries the NaN in channel 1 too, which keeps e0..e3 false by itself
                    const bool a0 = IL || p0.x == p0.x, a1 = IL || p0.y == p0.y, a2 = IL || p0.z == p0.z, a3 = IL || p0.w == p0.w;
                    if (__builtin_amdgcn_ballot_w64((e0 && a0) || (e1 && a1) || (e2 && a2) || (e3 && a3)) != 0) {
                        acc += ((e0 && a0) ? 0.0f : v0) + ((e1 && a1) ? 0.0f : v1)
                             + ((e2 && a2) ? 0.0f : v2) + ((e3 && a3) ? 0.0f : v3);
                        acc_edge += ((e0 && a0) ? v0 : 0.0f) + ((e1 && a1) ? v1 : 0.0f)
                                  + ((e2 && a2) ? v2 : 0.0f) + ((e3 && a3) ? v3 : 0.0f);
                        const bool ea[4] = {e0 && a0, e1 && a1, e2 && a2, e3 && a3};
                        for (int k = 0; k < 4; ++k)
                            if (ea[k]) {
                                const unsigned int slot = atomicAdd(KA(edge_count), 1u);
                                if (slot < KA(edge_cap)) KA(edges)[slot] = EdgeEntry{cell, l, 4 * q + k};
                            }
                    } else {
                        acc += (v0 + v1) + (v2 + v3);
                    }
                } else {
                    acc += (v0 + v1) + (v2 + v3);
                }
                return false;
            };
            // No register prefetch of the next group: at 5 waves per SIMD the other waves cover a group's load
            // latency (a group is ~430 VALU instructions), and the second register set was what kept the
            // kernel at 4 waves (155 VGPRs wanted, 128 allowed, spills).
            for (int step = 0; step < groups; ++step) {
                const int j = (sorted || sorted64) ? __builtin_amdgcn_readlane(order, step) : step;
                const LibGroup g = load_lib_group<TYPE, kNoQc>(b0, b1, b2, bc, IL ? lane + 4 * kWave * j : lane + kWave * j);
                if (visit(g, j, step)) { pruned = true; break; }
            }
        }
        if (lane == 0) {                        // the counters are per lane (stage 1 counts on the candidates' first lanes)
            n_visited += (unsigned int)visited;
            if (!two_level && !(BIG && stage0)) n_groups_all += (unsigned int)groups;      // (stage 0 counted its candidates itself)
            n_bound_dropped += (pruned && visited == 0) ? 1u : 0u;
        }
        acc = wave_sum_all(acc + acc_edge);
#if MM_WITH_PRETIER
        if constexpr (PRE) {
            if (pre_sum >= 0.0f && !pruned) pretier_canary(pre_sum, acc, lane, groups, MM_VAR.pre_abs, MM_VAR.pre_rel, KA(edge_count));
        }
#endif
        if (lane == 0) KA(scores)[(size_t)cell * KA(L) + l] = pruned ? __builtin_huge_valf() : acc;
    }
    MM_STAMP(5);
#ifdef MM_FILTER_STAMPS
    stamp_items += 1;
#endif
    }   // next item
#ifdef MM_FILTER_STAMPS
    __syncthreads();
    if (threadIdx.x < 6) atomicAdd(KA(edge_count) + 24 + threadIdx.x, stamp_acc[threadIdx.x] >> 6);
    if (threadIdx.x == 0 && stamp_items > 0) {      // 6: the longest workgroup lifetime, 7: the sum of the lifetimes (workgroups that took an item)
        const unsigned int life = (unsigned int)((__builtin_readcyclecounter() - stamp_begin) >> 10);
        atomicMax(KA(edge_count) + 30, life);
        atomicAdd(KA(edge_count) + 31, life >> 4);
    }
#endif
    {   // one set of atomics per wave and launch
        for (int off = 32; off > 0; off >>= 1) {
            n_visited += __shfl_xor(n_visited, off, kWave);
            n_groups_all += __shfl_xor(n_groups_all, off, kWave);
            n_bound_dropped += __shfl_xor(n_bound_dropped, off, kWave);
        }
        if ((threadIdx.x & 63) == 0) {
            unsigned long long* c = reinterpret_cast<unsigned long long*>(KA(edge_count) + kCounterBase);
            atomicAdd(c, (unsigned long long)n_visited);
            atomicAdd(c + 1, (unsigned long long)n_groups_all);
            atomicAdd(c + 2, (unsigned long long)n_bound_dropped);
        }
    }
}

#undef MM_VAR
#undef KA

// LDS-resident form of filter_kernel: a 1024-thread workgroup (16 waves) stages ONE cell
// -- its three colour planes, chroma plane and mask, only the groups holding active
// pixels -- in LDS once, then each wave walks its share of a 64-candidate tile against
// it.  Global loads are then the library planes only (16 B per pixel pair instead of 33 B
// with the cell coming back from L2 for every candidate), and each library group is
// fetched one iteration ahead of its use.  Same arithmetic, same exits, same results as
// filter_kernel; used whenever the cell fits LDS beside the hue table.
constexpr int kLdsBlock = 1024;

// (RGB / CIE76: two 16-wave workgroups per CU need 8 waves per SIMD, i.e. at most 64 VGPRs -- left alone the compiler takes 80
// and one workgroup per CU stays resident)
// CELL_GLOBAL (RGB / CIE76 only): cells whose planes do not fit LDS (128 x 128: 196 KB -- the reference GUI's default cell
// size at its default detail of 100 %) keep everything else of this kernel -- the cell's block sums in LDS, stage 1 with one
// candidate per lane, the survivor list -- and the few survivors' waves read the cell's pixels from global memory (L2: the
// sixteen waves of the workgroup walk the same 196 KB).
template <int TYPE, bool CELL_GLOBAL = false>
__global__ void __launch_bounds__(kLdsBlock) __attribute__((amdgpu_waves_per_eu(TYPE == CIEDE2000 ? 4 : MM_EUCLID_WAVES, TYPE == CIEDE2000 ? 4 : 8)))
filter_lds_kernel(const float* __restrict__ cells, const float* __restrict__ cells_f0,
                  const float* __restrict__ cell_chroma, const float* __restrict__ lib,
                  const float* __restrict__ lib_chroma, float* __restrict__ scores,
                  int n_cells, int L, int ppad, int cand_per_block, int cell_group,
                  EdgeEntry* __restrict__ edges, unsigned int* __restrict__ edge_count,
                  unsigned int edge_cap, const float* __restrict__ prune_above,
                  const HueEntry* __restrict__ hue_lut, Variants var, const float* __restrict__ lib_stats_t)
{
    // lib_stats_t (Euclid; may be null): the three channel sums per block, candidate-fastest: [variant][3][ppad / 64][L]
    extern __shared__ float4 smem[];
    __shared__ int n_surv;
    __shared__ unsigned int wg_counts[3];
    HueEntry* lut = reinterpret_cast<HueEntry*>(smem);              // [kHueLutEntries] (CIEDE2000), 16-byte multiple
    float4* s0 = smem + (TYPE == CIEDE2000 ? kHueLutEntries / 2 : 0);  // cell planes, ppad/4 float4 each
    const int pq = ppad / 4;
    float4* s1 = s0 + pq;
    float4* s2 = s1 + pq;
    float4* sc = s2 + pq;
    // Euclid with the early exit: behind the three planes the cell's block sums + weights [4][ppad / 64] and the tile's survivors
    float* cst = CELL_GLOBAL ? reinterpret_cast<float*>(smem) : reinterpret_cast<float*>(s2 + pq);
    int* surv = reinterpret_cast<int*>(cst + 4 * (ppad / 64));
    if (threadIdx.x < 3) wg_counts[threadIdx.x] = 0u;

    const int n_tiles = (L + cand_per_block - 1) / cand_per_block;
    const float* const lib_base = lib;
    const float* const lib_chroma_base = lib_chroma;
    unsigned int n_visited = 0, n_groups_all = 0, n_bound_dropped = 0;     // work counters (per lane; as filter_kernel's)
    // a grid may not exceed 2^32 threads: the launch is capped and the workgroups stride over the (cell, tile) items
    for (long long bid = blockIdx.x; bid < (long long)n_cells * n_tiles; bid += gridDim.x) {
    if (bid != (long long)blockIdx.x) __syncthreads();        // the previous item is done with the staged cell
    const long long per_group = (long long)cell_group * n_tiles;
    const int group = (int)(bid / per_group);
    const long long rem = bid - (long long)group * per_group;
    const int cells_here = min(cell_group, n_cells - group * cell_group);
    const int tile = (int)(rem / cells_here);
    const int cell = group * cell_group + (int)(rem - (long long)tile * cells_here);
    if (tile >= n_tiles || cell >= n_cells) continue;         // whole workgroup: uniform

    const int v = var.cell_var[cell];
    const int groups = var.groups[v];
    lib = lib_base + (size_t)v * var.lib_stride;
    lib_chroma = lib_chroma_base + (size_t)v * var.chroma_stride;

    // stage the hue table and the cell (active groups only)
    if constexpr (TYPE == CIEDE2000)
        for (int i = threadIdx.x; i < kHueLutEntries; i += kLdsBlock) lut[i] = hue_lut[i];
    const float4* const g0 = reinterpret_cast<const float4*>(cells + (size_t)cell * 3 * ppad);
    const float4* const gf = reinterpret_cast<const float4*>(cells_f0 + (size_t)cell * ppad);     // NaN = inactive
    if constexpr (!CELL_GLOBAL) {
        const float4* gc = reinterpret_cast<const float4*>(cell_chroma + (size_t)cell * ppad);
        const int nq = groups * kWave;
        for (int q = threadIdx.x; q < nq; q += kLdsBlock) {
            s0[q] = gf[q];
            s1[q] = g0[pq + q];
            s2[q] = g0[2 * pq + q];
            if constexpr (TYPE == CIEDE2000) sc[q] = gc[q];
        }
    }
    // Stage 1 (Euclid, early exit on): on the synthetic benchmark content the block bound ALONE drops 98 % of config B's
    // candidates, so it is evaluated one candidate per LANE -- each lane walks the blocks of its own candidate through the
    // candidate-fastest copy of the library's block sums (coalesced), ~12 instructions per block and candidate -- before any
    // wave is spent on a candidate; the few survivors go into an LDS list and get the wave-per-candidate treatment below.
    const int nb = ppad / 64;
    bool staged1 = false;
    if constexpr (TYPE != CIEDE2000) {
        // (cells above 64 groups, CELL_GLOBAL only: stage 1 has no size limit of its own -- the survivors then walk their pixels
        // with the plain partial-sum exit, the per-wave prelude of the rest bound holding at most 64 x 4 blocks)
        staged1 = lib_stats_t != nullptr && prune_above != nullptr && var.cell_full[cell] != 0 && (CELL_GLOBAL || groups <= kWave);
        if (staged1) {
            const bool clip1 = var.cell_full[cell] == 2;
            const float* ct = var.cell_stats + (size_t)cell * var.n_blocks;
            const size_t cs = var.cell_stats_stride;
            for (int t = threadIdx.x; t < 4 * groups; t += kLdsBlock) {
                cst[t] = ct[t]; cst[nb + t] = ct[2 * cs + t]; cst[2 * nb + t] = ct[3 * cs + t];
                cst[3 * nb + t] = clip1 ? ct[7 * cs + t] : 1.0f;
            }
            if (threadIdx.x == 0) n_surv = 0;
        }
    }
    __syncthreads();

    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const float limit = prune_above ? prune_above[cell] : __builtin_huge_valf();
    const int l_first = tile * cand_per_block;
    if constexpr (TYPE != CIEDE2000) if (staged1) {
        const int n_here = min(cand_per_block, L - l_first);
        const float* st_v = lib_stats_t + (size_t)v * 3 * nb * L;
        for (int idx = threadIdx.x; idx < n_here; idx += kLdsBlock) {
            const int l = l_first + idx;
            float b = 0.0f;
            // four blocks' sums in flight at a time (12 independent loads): the loop is latency, not arithmetic
            for (int blk0 = 0; blk0 < 4 * groups; blk0 += 4) {      // (4 * groups is a multiple of 4)
                float q0[4], q1[4], q2[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    q0[u] = st_v[(size_t)(blk0 + u) * L + l];
                    q1[u] = st_v[((size_t)nb + blk0 + u) * L + l];
                    q2[u] = st_v[((size_t)2 * nb + blk0 + u) * L + l];
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int blk = blk0 + u;
                    const float d0 = fmaxf(fabsf(cst[blk] - q0[u]) - 2e-2f, 0.0f);
                    const float d1 = fmaxf(fabsf(cst[nb + blk] - q1[u]) - 2e-2f, 0.0f);
                    const float d2 = fmaxf(fabsf(cst[2 * nb + blk] - q2[u]) - 2e-2f, 0.0f);
                    b += cst[3 * nb + blk] * 0.9999f * fsqrt(fmaf(d0, d0, fmaf(d1, d1, d2 * d2)));
                }
            }
            if (b * var.bound_scale > limit) {        // (bound_scale: 1; diagnostics)
                scores[(size_t)cell * L + l] = __builtin_huge_valf();
                n_bound_dropped += 1;
                n_groups_all += (unsigned int)groups;
            } else {
                surv[atomicAdd(&n_surv, 1)] = idx;
            }
        }
        __syncthreads();
    }
    // Euclidean differences: the lower bound on the unvisited pixels needs no constants at all --
    // by the triangle inequality of the Euclidean norm over a block's pixels
    //   sum_block sqrt(d0^2 + d1^2 + d2^2) >= sqrt((sum d0)^2 + (sum d1)^2 + (sum d2)^2),
    // from the three channel sums per image and 64-position block (group_sums_kernel).  Same
    // prelude as filter_kernel: lane k takes blocks k*m .., suffix scan, group j reads lane (4 j) >> shift.
    // Only for unclipped cells (the library's sums are over the mask's pixels).
    bool use_lb = false;
    if constexpr (TYPE != CIEDE2000) use_lb = prune_above != nullptr && var.cell_full[cell] != 0 && groups <= kWave;
    const bool clipped = var.cell_full[cell] == 2;
    const int lane_shift = groups <= 16 ? 0 : (groups <= 32 ? 1 : 2), per_lane = 1 << lane_shift;
    const int l_end = min((tile + 1) * cand_per_block, L);
    const int n_todo = staged1 ? n_surv : l_end - l_first;
    for (int turn = wave; turn < n_todo; turn += kLdsBlock / kWave) {
        const int l = l_first + (staged1 ? surv[turn] : turn);
        const float4* b0 = reinterpret_cast<const float4*>(lib + (size_t)l * 3 * ppad);
        const float4* b1 = b0 + pq;
        const float4* b2 = b1 + pq;
        const float4* bc = reinterpret_cast<const float4*>(lib_chroma + (size_t)l * ppad);
        bool pruned = false;
        float rest = 0.0f;                 // lane k: bound of everything from lane k's first block on
        if (use_lb) {
            const int n_blk = 4 * groups;
            float d = 0.0f;
            for (int i = 0; i < per_lane; ++i) {
                const int blk = lane * per_lane + i;
                if (blk >= n_blk) break;
                const float* st = var.lib_stats + ((size_t)v * L + l) * var.n_blocks + blk;
                const float* ct = var.cell_stats + (size_t)cell * var.n_blocks + blk;
                const size_t bs = var.lib_stats_stride, cs = var.cell_stats_stride;
                // 2e-2: fp32 rounding of two 64-term sums of values <= 255, with margin
                const float d0 = fmaxf(fabsf(ct[0] - st[0]) - 2e-2f, 0.0f);
                const float d1 = fmaxf(fabsf(ct[2 * cs] - st[2 * bs]) - 2e-2f, 0.0f);
                const float d2 = fmaxf(fabsf(ct[3 * cs] - st[3 * bs]) - 2e-2f, 0.0f);
                // (a clipped cell: only blocks whose mask pixels are all inside its bound -- statistic 7 -- count)
                d += (clipped ? ct[7 * cs] : 1.0f) * 0.9999f * fsqrt(fmaf(d0, d0, fmaf(d1, d1, d2 * d2)));
            }
            for (int off = 1; off < kWave; off <<= 1) {       // inclusive suffix scan over lanes
                const float up = __shfl_down(d, off, kWave);
                if (lane + off < kWave) d += up;
            }
            rest = d * var.bound_scale;
            pruned = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, rest), 0)) > limit;
        }
        float acc = 0.0f, acc_edge = 0.0f;
        int visited = 0;
        if (!pruned && groups > 0) {
            // (the library group one iteration ahead of its use -- CIEDE2000 only: the Euclidean form is held to 64 VGPRs for
            // two workgroups per CU, and the second register set is what would spill)
            constexpr bool kAhead = TYPE == CIEDE2000 || MM_EUCLID_PREFETCH;
            LibGroup nxt = load_lib_group<TYPE>(b0, b1, b2, bc, lane);
            for (int j = 0; j < groups; ++j) {
                if (j != 0) {
                    const float rest_j = use_lb ? __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, rest), (4 * j) >> lane_shift)) : 0.0f;
                    if (wave_sum_exceeds(acc, rest_j, limit)) { pruned = true; break; }
                }
                visited += 1;
                const int q = lane + kWave * j;
                const LibGroup cur = (kAhead || j == 0) ? nxt : load_lib_group<TYPE>(b0, b1, b2, bc, q);
                if (kAhead && j + 1 < groups) nxt = load_lib_group<TYPE>(b0, b1, b2, bc, q + kWave);   // one group ahead
                const float4 p0 = CELL_GLOBAL ? gf[q] : s0[q], p1 = CELL_GLOBAL ? g0[pq + q] : s1[q], p2 = CELL_GLOBAL ? g0[2 * pq + q] : s2[q];
                float4 pc{0.f, 0.f, 0.f, 0.f};
                if constexpr (TYPE == CIEDE2000) pc = sc[q];
                const float4 q0 = cur.q0, q1 = cur.q1, q2 = cur.q2, qc = cur.qc;
                bool e0, e1, e2, e3;
                const float v0 = pair_fast<TYPE>(PxF{p0.x, p1.x, p2.x}, PxF{q0.x, q1.x, q2.x}, pc.x, qc.x, lut, &e0);
                const float v1 = pair_fast<TYPE>(PxF{p0.y, p1.y, p2.y}, PxF{q0.y, q1.y, q2.y}, pc.y, qc.y, lut, &e1);
                const float v2 = pair_fast<TYPE>(PxF{p0.z, p1.z, p2.z}, PxF{q0.z, q1.z, q2.z}, pc.z, qc.z, lut, &e2);
                const float v3 = pair_fast<TYPE>(PxF{p0.w, p1.w, p2.w}, PxF{q0.w, q1.w, q2.w}, pc.w, qc.w, lut, &e3);
                if constexpr (TYPE == CIEDE2000) {
                    const bool a0 = p0.x == p0.x, a1 = p0.y == p0.y, a2 = p0.z == p0.z, a3 = p0.w == p0.w;
                    if (__builtin_amdgcn_ballot_w64((e0 && a0) || (e1 && a1) || (e2 && a2) || (e3 && a3)) != 0) {
                        acc += ((e0 && a0) ? 0.0f : v0) + ((e1 && a1) ? 0.0f : v1)
                             + ((e2 && a2) ? 0.0f : v2) + ((e3 && a3) ? 0.0f : v3);
                        acc_edge += ((e0 && a0) ? v0 : 0.0f) + ((e1 && a1) ? v1 : 0.0f)
                                  + ((e2 && a2) ? v2 : 0.0f) + ((e3 && a3) ? v3 : 0.0f);
                        const bool ea[4] = {e0 && a0, e1 && a1, e2 && a2, e3 && a3};
                        for (int k = 0; k < 4; ++k)
                            if (ea[k]) {
                                const unsigned int slot = atomicAdd(edge_count, 1u);
                                if (slot < edge_cap) edges[slot] = EdgeEntry{cell, l, 4 * q + k};
                            }
                    } else {
                        acc += (v0 + v1) + (v2 + v3);
                    }
                } else {
                    acc += (v0 + v1) + (v2 + v3);
                }
            }
        }
        acc = wave_sum_f(acc + acc_edge);
        if (lane == 0) {
            scores[(size_t)cell * L + l] = pruned ? __builtin_huge_valf() : acc;
            n_visited += (unsigned int)visited;
            n_groups_all += (unsigned int)groups;
            n_bound_dropped += (pruned && visited == 0) ? 1u : 0u;
        }
    }
    }
    // work counters: summed over the workgroup in LDS, then ONE set of global atomics per workgroup (a set per wave is
    // 10^5 atomics on three addresses in a 0.3 ms launch: it quadrupled config B's kernel time when it was tried)
    for (int off = 32; off > 0; off >>= 1) {
        n_visited += __shfl_xor(n_visited, off, kWave);
        n_groups_all += __shfl_xor(n_groups_all, off, kWave);
        n_bound_dropped += __shfl_xor(n_bound_dropped, off, kWave);
    }
    __syncthreads();
    if ((threadIdx.x & 63) == 0) {
        atomicAdd(&wg_counts[0], n_visited); atomicAdd(&wg_counts[1], n_groups_all); atomicAdd(&wg_counts[2], n_bound_dropped);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long* c = reinterpret_cast<unsigned long long*>(edge_count + kCounterBase);
        atomicAdd(c, (unsigned long long)wg_counts[0]);
        atomicAdd(c + 1, (unsigned long long)wg_counts[1]);
        atomicAdd(c + 2, (unsigned long long)wg_counts[2]);
    }
}

// One pair for the seed scores: fp32 form, with the fp64 form taken inline on the
// hue discontinuity (the seeds are few, the divergence is irrelevant there).
template <int TYPE>
__device__ __forceinline__ float pair_seed(const PxF& p, const PxF& q, float cp, float cq, const HueEntry* lut)
{
    bool edge;
    float v = pair_fast<TYPE>(p, q, cp, cq, lut, &edge);
    if constexpr (TYPE == CIEDE2000)
        if (edge) v = (float)ciede2000(Px{p.c0, p.c1, p.c2}, Px{q.c0, q.c1, q.c2});
    return v;
}

// fp32 filter score of LISTED candidates (the seeds whose scores bound the pruning
// limit): one wave per (cell, slot), no early exit; same values as filter_kernel +
// edge_fix_kernel produce, up to fp32 summation order.
template <int TYPE>
__global__ void __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(MM_SEED_WAVES, MM_SEED_WAVES)))
seed_score_kernel(const float* __restrict__ cells, const uint8_t* __restrict__ cmask,
                  const float* __restrict__ cell_chroma, const float* __restrict__ lib,
                  const float* __restrict__ lib_chroma, const int32_t* __restrict__ seeds,
                  float* __restrict__ seed_score, int n_cells, int M, int ppad,
                  const HueEntry* __restrict__ hue_lut, Variants var, float* __restrict__ scores, int L)
{
    __shared__ HueEntry lut[TYPE == CIEDE2000 ? kHueLutEntries : 1];
    if constexpr (TYPE == CIEDE2000) load_hue_lut(lut, hue_lut);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const float* const lib_base = lib;
    const float* const lib_chroma_base = lib_chroma;
    // grid-stride over the (cell, slot) pairs: the hue table is loaded once per workgroup, not once per four pairs
    for (long long w = (long long)blockIdx.x * (kBlock / kWave) + wave; w < (long long)n_cells * M;
         w += (long long)gridDim.x * (kBlock / kWave)) {
    const int cell = (int)(w / M);
    const int l = seeds[w];
    if (l < 0) { if (lane == 0) seed_score[w] = __builtin_huge_valf(); continue; }
    const int v = var.cell_var[cell];
    const int groups = var.groups[v];
    lib = lib_base + (size_t)v * var.lib_stride;
    lib_chroma = lib_chroma_base + (size_t)v * var.chroma_stride;
    const float4* c0 = reinterpret_cast<const float4*>(cells + (size_t)cell * 3 * ppad);
    const float4* c1 = c0 + ppad / 4;
    const float4* c2 = c1 + ppad / 4;
    const uint32_t* mk = reinterpret_cast<const uint32_t*>(cmask + (size_t)cell * ppad);
    const float4* b0 = reinterpret_cast<const float4*>(lib + (size_t)l * 3 * ppad);
    const float4* b1 = b0 + ppad / 4;
    const float4* b2 = b1 + ppad / 4;
    const float4* cc = reinterpret_cast<const float4*>(cell_chroma + (size_t)cell * ppad);
    const float4* bc = reinterpret_cast<const float4*>(lib_chroma + (size_t)l * ppad);
    float acc = 0.0f;
    for (int j = 0; j < groups; ++j) {
        const int q = lane + kWave * j;
        const uint32_t m = mk[q];
        if (m == 0) continue;
        const float4 p0 = c0[q], p1 = c1[q], p2 = c2[q];
        const float4 q0 = b0[q], q1 = b1[q], q2 = b2[q];
        float4 pc{0.f, 0.f, 0.f, 0.f}, qc{0.f, 0.f, 0.f, 0.f};
        if constexpr (TYPE == CIEDE2000) { pc = cc[q]; qc = bc[q]; }
        const float v0 = pair_seed<TYPE>(PxF{p0.x, p1.x, p2.x}, PxF{q0.x, q1.x, q2.x}, pc.x, qc.x, lut);
        const float v1 = pair_seed<TYPE>(PxF{p0.y, p1.y, p2.y}, PxF{q0.y, q1.y, q2.y}, pc.y, qc.y, lut);
        const float v2 = pair_seed<TYPE>(PxF{p0.z, p1.z, p2.z}, PxF{q0.z, q1.z, q2.z}, pc.z, qc.z, lut);
        const float v3 = pair_seed<TYPE>(PxF{p0.w, p1.w, p2.w}, PxF{q0.w, q1.w, q2.w}, pc.w, qc.w, lut);
        acc += ((m & 0x000000ffu) ? v0 : 0.0f) + ((m & 0x0000ff00u) ? v1 : 0.0f)
             + ((m & 0x00ff0000u) ? v2 : 0.0f) + ((m & 0xff000000u) ? v3 : 0.0f);
    }
    acc = wave_sum_f(acc);
    if (lane == 0) {
        seed_score[w] = acc;
        // the seed IS a filter score of this (cell, candidate): deposited in the score row; filter_kernel looks the
        // candidate up in the cell's seed list and leaves it alone instead of scoring it a second time
        if (scores) scores[(size_t)cell * L + l] = acc;
    }
    }
}

// Chroma sqrt(a^2 + b^2) of every pixel of n planar images: [n][3][ppad] -> [n][ppad].
__global__ void __launch_bounds__(kBlock)
chroma_planes_kernel(const float* __restrict__ planar, float* __restrict__ chroma, size_t n_images, int ppad)
{
    const size_t total = n_images * (size_t)(ppad / 4);
    for (size_t t = (size_t)blockIdx.x * kBlock + threadIdx.x; t < total; t += (size_t)gridDim.x * kBlock) {
        const size_t img = t / (size_t)(ppad / 4);
        const size_t q = t - img * (size_t)(ppad / 4);
        const float4* base = reinterpret_cast<const float4*>(planar + img * 3 * (size_t)ppad);
        const float4 a = base[ppad / 4 + q], b = base[2 * (size_t)(ppad / 4) + q];
        reinterpret_cast<float4*>(chroma + img * (size_t)ppad)[q] =
            float4{chroma_of(a.x, b.x), chroma_of(a.y, b.y), chroma_of(a.z, b.z), chroma_of(a.w, b.w)};
    }
}

// Interleaved copy of n planar images for filter_kernel<IL>: [n][3][ppad] (+ chroma [n][ppad], + an optional
// replacement of channel 0, e.g. the NaN-marked one of the cells) -> [n][ppad / 256][4][256].
__global__ void __launch_bounds__(kBlock)
interleave_kernel(const float* __restrict__ planar, const float* __restrict__ chroma, const float* __restrict__ plane0,
                  float* __restrict__ out, size_t n_images, int ppad)
{
    const size_t per = (size_t)(ppad / 4), total = n_images * per;
    for (size_t t = (size_t)blockIdx.x * kBlock + threadIdx.x; t < total; t += (size_t)gridDim.x * kBlock) {
        const size_t img = t / per;
        const int q = (int)(t - img * per);                       // float4 position in a plane
        const float4* src = reinterpret_cast<const float4*>(planar + img * 3 * (size_t)ppad);
        float4* dst = reinterpret_cast<float4*>(out + img * 4 * (size_t)ppad) + (size_t)(q >> 6) * 4 * kWave + (q & (kWave - 1));
        float4 first = plane0 ? reinterpret_cast<const float4*>(plane0 + img * (size_t)ppad)[q] : src[q];
        float4 second = src[per + q];
        // a NaN in the replacement of channel 0 marks an inactive pixel: channel 1 carries the mark too, so that every
        // quantity of the pair is NaN there (no "on the discontinuity" verdict for a pixel that does not count)
        if (first.x != first.x) second.x = first.x;
        if (first.y != first.y) second.y = first.y;
        if (first.z != first.z) second.z = first.z;
        if (first.w != first.w) second.w = first.w;
        // the prepared operands of ciede2000_fast<PREP = true>: L / 2 - 25 and C / 50
        first = float4{fmaf(0.5f, first.x, -25.0f), fmaf(0.5f, first.y, -25.0f), fmaf(0.5f, first.z, -25.0f), fmaf(0.5f, first.w, -25.0f)};
        dst[0] = first;
        dst[kWave] = second;
        dst[2 * kWave] = src[2 * per + q];
        float4 ch = chroma ? reinterpret_cast<const float4*>(chroma + img * (size_t)ppad)[q] : float4{0.f, 0.f, 0.f, 0.f};
        dst[3 * kWave] = float4{__fmul_rn(kChromaPrep, ch.x), __fmul_rn(kChromaPrep, ch.y), __fmul_rn(kChromaPrep, ch.z), __fmul_rn(kChromaPrep, ch.w)};
    }
}

#if MM_WITH_PRETIER
// The interleaved copy once more as halfs, for the packed-fp16 pre-tier (colour_diff_fast.hpp ciede2000_h2): the same
// [image][group][plane][256] order, planes L / 2 - 25, a / 4, b / 4, C / 4, round-to-nearest; a NaN stays a NaN.
__global__ void __launch_bounds__(kBlock)
interleave_h_kernel(const float4* __restrict__ il, uint2* __restrict__ out, size_t n_vec)
{
    for (size_t t = (size_t)blockIdx.x * kBlock + threadIdx.x; t < n_vec; t += (size_t)gridDim.x * kBlock) {
        const int plane = (int)((t >> 6) & 3);                    // 64 float4 per plane, four planes per group
        const float sc = (plane == 1 || plane == 2) ? kHalfOperandScale : plane == 3 ? kHalfChromaScale : 1.0f;
        const float4 v = il[t];
        const half2_t lo{(half1_t)(sc * v.x), (half1_t)(sc * v.y)}, hi{(half1_t)(sc * v.z), (half1_t)(sc * v.w)};
        out[t] = uint2{h2_bits(lo), h2_bits(hi)};
    }
}
#endif

// Per-image statistics over each 64-position BLOCK of the plane order (four blocks make one of
// the filter's 256-position groups; the plane order is tile-major, engine.hip prepare_planes, so
// a block is a compact patch of the image), counting only the positions flagged in `mask`
// (per image if mask_stride != 0, else shared).  Seven arrays of [n_images][n_blocks] floats,
// `block_stride` floats apart:
//   0: sum of channel 0              1: max |channel 0 - 50|
//   2: sum of channel 1              3: sum of channel 2        4: max sqrt(ch1^2 + ch2^2)
//   5, 6: the arc of the hue circle holding every active pixel's hue: centre as an offset from 275 deg in
//         [-180, 180], and half-width, in degrees.  A pixel's hue is that of (g a, b) for every g = 1 + G in
//         [1, 1.5] (ColourDifference.cpp:50-53,61-83): both ends count.  Deviations are measured from the
//         direction of the block's summed (a, b), so no fixed seam cuts a block in two; widened by 0.02 deg;
//         the whole circle (centre 0, half-width 180) when the sum vanishes, a pixel's chroma is below 1e-3
//         (its hue is arbitrary and the reference's mean-hue rule changes, :108-109), a pixel lies more than
//         170 deg off the sum, or the block has no active pixel.
// One wave per (image, group), sixteen lanes per block.  Feeds the lower bound of filter_kernel's early exit.
constexpr int LPB = 16;
__global__ void __launch_bounds__(kBlock)
group_sums_kernel(const float* __restrict__ planar, const uint8_t* __restrict__ mask, size_t mask_stride,
                  float* __restrict__ stats, size_t block_stride, size_t n_images, int n_groups,
                  int n_blocks, int ppad)
{
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const size_t w = (size_t)blockIdx.x * (kBlock / kWave) + wave;
    if (w >= n_images * (size_t)n_groups) return;
    const size_t img = w / n_groups;
    const int g = (int)(w - img * n_groups);
    const int q = g * kWave + lane;
    const float4* base = reinterpret_cast<const float4*>(planar + img * 3 * (size_t)ppad);
    const float4 x = base[q], a = base[ppad / 4 + q], b = base[2 * (size_t)(ppad / 4) + q];
    const uint32_t m = reinterpret_cast<const uint32_t*>(mask + img * mask_stride)[q];
    const bool on[4] = {(m & 0x000000ffu) != 0, (m & 0x0000ff00u) != 0, (m & 0x00ff0000u) != 0, (m & 0xff000000u) != 0};
    const float xs[4] = {x.x, x.y, x.z, x.w}, as[4] = {a.x, a.y, a.z, a.w}, bs[4] = {b.x, b.y, b.z, b.w};
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, dev = 0.f, cmax = 0.f;
    bool anywhere = false, any = false;
    for (int k = 0; k < 4; ++k)
        if (on[k]) {
            s0 += xs[k]; s1 += as[k]; s2 += bs[k];
            dev = fmaxf(dev, fabsf(xs[k] - 50.0f));
            const float ch = chroma_of(as[k], bs[k]);
            cmax = fmaxf(cmax, ch);
            anywhere = anywhere || !(ch >= 1e-3f);
            any = true;
        }
    for (int off = LPB / 2; off > 0; off >>= 1) {        // within the block's lanes
        s0 += __shfl_xor(s0, off, kWave); s1 += __shfl_xor(s1, off, kWave); s2 += __shfl_xor(s2, off, kWave);
        dev = fmaxf(dev, __shfl_xor(dev, off, kWave));
        cmax = fmaxf(cmax, __shfl_xor(cmax, off, kWave));
        anywhere = anywhere || (__shfl_xor((int)anywhere, off, kWave) != 0);
        any = any || (__shfl_xor((int)any, off, kWave) != 0);
    }
    // second pass: every pixel's hue (both ends of its g-range) against the direction of the block's sum
    const float ref = atan2f(s2, s1) * 57.29577951308232f;
    float dlo = 180.0f, dhi = -180.0f, far = 0.0f;
    for (int k = 0; k < 4; ++k)
        if (on[k]) {
            const float e1 = wrap180(atan2f(bs[k], as[k]) * 57.29577951308232f - ref);
            const float e2 = wrap180(atan2f(bs[k], 1.5f * as[k]) * 57.29577951308232f - ref);
            dlo = fminf(dlo, fminf(e1, e2)); dhi = fmaxf(dhi, fmaxf(e1, e2));
            far = fmaxf(far, fmaxf(fabsf(e1), fabsf(e2)));
        }
    for (int off = LPB / 2; off > 0; off >>= 1) {
        dlo = fminf(dlo, __shfl_xor(dlo, off, kWave)); dhi = fmaxf(dhi, __shfl_xor(dhi, off, kWave));
        far = fmaxf(far, __shfl_xor(far, off, kWave));
    }
    if ((lane & (LPB - 1)) == 0) {
        float* o = stats + img * n_blocks + (kWave / LPB) * g + lane / LPB;
        o[0] = s0; o[block_stride] = dev; o[2 * block_stride] = s1; o[3 * block_stride] = s2; o[4 * block_stride] = cmax;
        const bool open = anywhere || !any || !(far <= 170.0f) || !(fmaf(s1, s1, s2 * s2) >= 1e-6f);
        o[5 * block_stride] = open ? 0.0f : wrap180(ref + 0.5f * (dlo + dhi) - 275.0f);
        o[6 * block_stride] = open ? 180.0f : 0.5f * (dhi - dlo) + 0.02f;
    }
}

// Whole-image statistics from the block statistics (see launch.hpp launch_image_stats): one WAVE per image -- lane b holds
// block b, b + 64, ... (coalesced) --, written into the image's own slot `image_slot` of the same seven arrays.
__global__ void __launch_bounds__(kBlock)
image_stats_kernel(float* __restrict__ stats, size_t block_stride, int slots_per_image, int image_slot, int active_blocks, size_t n_images,
                   const uint8_t* __restrict__ var_of_image, int4 blocks_of_var)
{
    const int lane = threadIdx.x & 63;
    const size_t img = (size_t)blockIdx.x * (kBlock / kWave) + (threadIdx.x >> 6);
    if (img >= n_images) return;
    if (var_of_image) {                 // a cell's statistics cover the blocks of ITS mask variant
        const int v = var_of_image[img];
        active_blocks = v == 0 ? blocks_of_var.x : v == 1 ? blocks_of_var.y : v == 2 ? blocks_of_var.z : blocks_of_var.w;
    }
    float* st = stats + img * slots_per_image;
    double s0 = 0.0, s1 = 0.0, s2 = 0.0;
    float dev = 0.f, cmax = 0.f;
    for (int b = lane; b < active_blocks; b += kWave) {
        s0 += (double)st[b]; s1 += (double)st[2 * block_stride + b]; s2 += (double)st[3 * block_stride + b];
        dev = fmaxf(dev, st[block_stride + b]);
        cmax = fmaxf(cmax, st[4 * block_stride + b]);
    }
    for (int off = 32; off > 0; off >>= 1) {
        s0 += __shfl_xor(s0, off, kWave); s1 += __shfl_xor(s1, off, kWave); s2 += __shfl_xor(s2, off, kWave);
        dev = fmaxf(dev, __shfl_xor(dev, off, kWave)); cmax = fmaxf(cmax, __shfl_xor(cmax, off, kWave));
    }
    // the blocks' hue arcs (centre as an offset from 275 deg, half-width) joined around the direction of the summed (a, b)
    const float ref = atan2f((float)s2, (float)s1) * 57.29577951308232f;
    float lo = 180.0f, hi = -180.0f;
    bool open = false;
    for (int b = lane; b < active_blocks; b += kWave) {
        const float c = st[5 * block_stride + b], w = st[6 * block_stride + b];
        const float off = wrap180(c + 275.0f - ref);
        open = open || !(fabsf(off) + w <= 170.0f);
        lo = fminf(lo, off - w); hi = fmaxf(hi, off + w);
    }
    for (int off = 32; off > 0; off >>= 1) { lo = fminf(lo, __shfl_xor(lo, off, kWave)); hi = fmaxf(hi, __shfl_xor(hi, off, kWave)); }
    open = __builtin_amdgcn_ballot_w64(open) != 0ull || active_blocks <= 0 || !(s1 * s1 + s2 * s2 >= 1e-6);
    if (lane == 0) {
        float* o = st + image_slot;
        o[0] = (float)s0; o[block_stride] = dev; o[2 * block_stride] = (float)s1; o[3 * block_stride] = (float)s2; o[4 * block_stride] = cmax;
        o[5 * block_stride] = open ? 0.0f : wrap180(ref + 0.5f * (lo + hi) - 275.0f);
        o[6 * block_stride] = open ? 180.0f : 0.5f * (hi - lo) + 0.02f;
    }
}

// Candidate-fastest copy of the three channel sums per block (statistics 0, 2, 3) for filter_lds_kernel's stage 1:
// out[(k * n_blocks + blk) * n_images + img].
__global__ void __launch_bounds__(kBlock)
transpose_stats_kernel(const float* __restrict__ stats, size_t block_stride, int slots_per_image, int n_blocks,
                       float* __restrict__ out, size_t n_images)
{
    const size_t total = (size_t)3 * n_blocks * n_images;
    for (size_t t = (size_t)blockIdx.x * kBlock + threadIdx.x; t < total; t += (size_t)gridDim.x * kBlock) {
        const size_t img = t % n_images;
        const size_t r = t / n_images;
        const int blk = (int)(r % n_blocks), k = (int)(r / n_blocks);
        const int stat = k == 0 ? 0 : k + 1;                      // 0, 2, 3
        out[t] = stats[(size_t)stat * block_stride + img * slots_per_image + blk];
    }
}

// Mean colour of every library image (over all pixels), f32 [L][3]: a proxy used only
// to pick which candidates to score first.
__global__ void __launch_bounds__(kBlock)
image_means_kernel(const float* __restrict__ planar, float* __restrict__ means, int P, int ppad)
{
    __shared__ float sh[3][kBlock / kWave];
    const float* img = planar + (size_t)blockIdx.x * 3 * ppad;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f;
    for (int p = threadIdx.x; p < P; p += kBlock) { s0 += img[p]; s1 += img[ppad + p]; s2 += img[2 * (size_t)ppad + p]; }
    s0 = wave_sum_f(s0); s1 = wave_sum_f(s1); s2 = wave_sum_f(s2);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (lane == 0) { sh[0][wave] = s0; sh[1][wave] = s1; sh[2][wave] = s2; }
    __syncthreads();
    if (threadIdx.x < 3) {
        float t = 0.f;
        for (int w = 0; w < kBlock / kWave; ++w) t += sh[threadIdx.x][w];
        means[(size_t)blockIdx.x * 3 + threadIdx.x] = t / (float)P;
    }
}

// Seeds of one cell: the M library images whose mean colour is closest (same fp32
// difference form) to the cell's mean colour over its active pixels.  One workgroup
// per cell, M rounds of a block-wide arg-min over L proxy distances kept in LDS.
template <int TYPE>
__global__ void __launch_bounds__(kBlock)
seed_kernel(const float* __restrict__ cells, const uint8_t* __restrict__ cmask,
            const float* __restrict__ lib_means, int L, int ppad, int M,
            int32_t* __restrict__ seeds, const HueEntry* __restrict__ hue_lut)
{
    __shared__ HueEntry lut[TYPE == CIEDE2000 ? kHueLutEntries : 1];
    if constexpr (TYPE == CIEDE2000) load_hue_lut(lut, hue_lut);
    __shared__ float red[4][kBlock / kWave];
    __shared__ float best_v[kBlock / kWave];
    __shared__ int best_i[kBlock / kWave];
    // every thread keeps the two closest of its stripe of the library (l = t, t + 256, ...): the M seeds are then
    // the M closest of those 512.  Exact up to 512 images; beyond, a third member of the true top M in one stripe
    // is passed over for the next best -- ANY M distinct candidates give a valid limit (the K-th lowest of their
    // true scores), the choice only has to be deterministic.  No table of L proxies: any library size.
    __shared__ float cand_v[2 * kBlock];
    __shared__ int cand_i[2 * kBlock];
    const float* c = cells + (size_t)blockIdx.x * 3 * ppad;
    const uint8_t* mk = cmask + (size_t)blockIdx.x * ppad;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, n = 0.f;
    for (int p = threadIdx.x; p < ppad; p += kBlock)
        if (mk[p]) { s0 += c[p]; s1 += c[ppad + p]; s2 += c[2 * (size_t)ppad + p]; n += 1.f; }
    s0 = wave_sum_f(s0); s1 = wave_sum_f(s1); s2 = wave_sum_f(s2); n = wave_sum_f(n);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (lane == 0) { red[0][wave] = s0; red[1][wave] = s1; red[2][wave] = s2; red[3][wave] = n; }
    __syncthreads();
    float t[4];
    for (int k = 0; k < 4; ++k) { t[k] = 0.f; for (int w = 0; w < kBlock / kWave; ++w) t[k] += red[k][w]; }
    const float inv = 1.0f / fmaxf(t[3], 1.0f);
    const PxF mean{t[0] * inv, t[1] * inv, t[2] * inv};
    const float mean_c = chroma_of(mean.c1, mean.c2);
    float v0 = __builtin_huge_valf(), v1 = __builtin_huge_valf();
    int i0 = 0x7fffffff, i1 = 0x7fffffff;
    for (int l = threadIdx.x; l < L; l += kBlock) {
        bool e;
        const PxF lm{lib_means[3 * l], lib_means[3 * l + 1], lib_means[3 * l + 2]};
        const float d = pair_fast<TYPE>(mean, lm, mean_c, chroma_of(lm.c1, lm.c2), lut, &e);
        if (d < v0) { v1 = v0; i1 = i0; v0 = d; i0 = l; }          // strict: the lower index stays ahead among equals
        else if (d < v1) { v1 = d; i1 = l; }
    }
    cand_v[threadIdx.x] = v0; cand_i[threadIdx.x] = i0;
    cand_v[kBlock + threadIdx.x] = v1; cand_i[kBlock + threadIdx.x] = i1;
    __syncthreads();
    for (int r = 0; r < M; ++r) {
        float v = cand_v[threadIdx.x];
        int i = cand_i[threadIdx.x];
        {
            const float ov = cand_v[kBlock + threadIdx.x];
            const int oi = cand_i[kBlock + threadIdx.x];
            if (ov < v || (ov == v && oi < i)) { v = ov; i = oi; }
        }
        for (int off = 32; off > 0; off >>= 1) {
            const float ov = __shfl_xor(v, off, kWave);
            const int oi = __shfl_xor(i, off, kWave);
            if (ov < v || (ov == v && oi < i)) { v = ov; i = oi; }
        }
        if (lane == 0) { best_v[wave] = v; best_i[wave] = i; }
        __syncthreads();
        if (threadIdx.x == 0) {
            for (int w = 1; w < kBlock / kWave; ++w)
                if (best_v[w] < v || (best_v[w] == v && best_i[w] < i)) { v = best_v[w]; i = best_i[w]; }
            seeds[(size_t)blockIdx.x * M + r] = (i == 0x7fffffff) ? -1 : i;
            if (i != 0x7fffffff) {                                  // taken: its slot is the one of thread i % 256 that holds i
                const int tq = i % kBlock;
                if (cand_i[tq] == i) { cand_v[tq] = __builtin_huge_valf(); cand_i[tq] = 0x7fffffff; }
                else { cand_v[kBlock + tq] = __builtin_huge_valf(); cand_i[kBlock + tq] = 0x7fffffff; }
            }
        }
        __syncthreads();
    }
}

// seed_kernel + seed_score_kernel + prune_limit_kernel in ONE launch, one workgroup per cell, for batches small
// enough that three dependent launches cost more than the balance they buy (the separate seed_score_kernel spreads
// the (cell, seed) pairs over the whole device at its own occupancy; config D keeps it).  Same seeds (the same
// selection code), the same per-seed arithmetic and lane assignment as seed_score_kernel, the same limit formula.
template <int TYPE>
__global__ void __launch_bounds__(kBlock)
seed_fused_kernel(const float* __restrict__ cells, const uint8_t* __restrict__ cmask,
                  const float* __restrict__ cell_chroma, const float* __restrict__ lib,
                  const float* __restrict__ lib_chroma, const float* __restrict__ lib_means,
                  int L, int ppad, int M, int K, float px_abs, float rel,
                  int32_t* __restrict__ seeds, float* __restrict__ seed_score, float* __restrict__ limit,
                  float* __restrict__ scores, const HueEntry* __restrict__ hue_lut, Variants var)
{
    __shared__ HueEntry lut[TYPE == CIEDE2000 ? kHueLutEntries : 1];
    if constexpr (TYPE == CIEDE2000) load_hue_lut(lut, hue_lut);
    __shared__ float red[4][kBlock / kWave];
    __shared__ float best_v[kBlock / kWave];
    __shared__ int best_i[kBlock / kWave];
    __shared__ float cand_v[2 * kBlock];
    __shared__ int cand_i[2 * kBlock];
    __shared__ int picked[256];              // the M seeds
    __shared__ float picked_score[256];
    const int cell = blockIdx.x;
    const float* c = cells + (size_t)cell * 3 * ppad;
    const uint8_t* mkb = cmask + (size_t)cell * ppad;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, n = 0.f;
    for (int p = threadIdx.x; p < ppad; p += kBlock)
        if (mkb[p]) { s0 += c[p]; s1 += c[ppad + p]; s2 += c[2 * (size_t)ppad + p]; n += 1.f; }
    s0 = wave_sum_f(s0); s1 = wave_sum_f(s1); s2 = wave_sum_f(s2); n = wave_sum_f(n);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (lane == 0) { red[0][wave] = s0; red[1][wave] = s1; red[2][wave] = s2; red[3][wave] = n; }
    __syncthreads();
    float t[4];
    for (int k = 0; k < 4; ++k) { t[k] = 0.f; for (int w = 0; w < kBlock / kWave; ++w) t[k] += red[k][w]; }
    const float n_active = t[3];
    const float inv = 1.0f / fmaxf(t[3], 1.0f);
    const PxF mean{t[0] * inv, t[1] * inv, t[2] * inv};
    const float mean_c = chroma_of(mean.c1, mean.c2);
    float v0 = __builtin_huge_valf(), v1 = __builtin_huge_valf();
    int i0 = 0x7fffffff, i1 = 0x7fffffff;
    for (int l = threadIdx.x; l < L; l += kBlock) {
        bool e;
        const PxF lm{lib_means[3 * l], lib_means[3 * l + 1], lib_means[3 * l + 2]};
        const float d = pair_fast<TYPE>(mean, lm, mean_c, chroma_of(lm.c1, lm.c2), lut, &e);
        if (d < v0) { v1 = v0; i1 = i0; v0 = d; i0 = l; }
        else if (d < v1) { v1 = d; i1 = l; }
    }
    cand_v[threadIdx.x] = v0; cand_i[threadIdx.x] = i0;
    cand_v[kBlock + threadIdx.x] = v1; cand_i[kBlock + threadIdx.x] = i1;
    __syncthreads();
    for (int r = 0; r < M; ++r) {
        float v = cand_v[threadIdx.x];
        int i = cand_i[threadIdx.x];
        {
            const float ov = cand_v[kBlock + threadIdx.x];
            const int oi = cand_i[kBlock + threadIdx.x];
            if (ov < v || (ov == v && oi < i)) { v = ov; i = oi; }
        }
        for (int off = 32; off > 0; off >>= 1) {
            const float ov = __shfl_xor(v, off, kWave);
            const int oi = __shfl_xor(i, off, kWave);
            if (ov < v || (ov == v && oi < i)) { v = ov; i = oi; }
        }
        if (lane == 0) { best_v[wave] = v; best_i[wave] = i; }
        __syncthreads();
        if (threadIdx.x == 0) {
            for (int w = 1; w < kBlock / kWave; ++w)
                if (best_v[w] < v || (best_v[w] == v && best_i[w] < i)) { v = best_v[w]; i = best_i[w]; }
            const int id = (i == 0x7fffffff) ? -1 : i;
            seeds[(size_t)cell * M + r] = id;
            picked[r] = id;
            if (i != 0x7fffffff) {
                const int tq = i % kBlock;
                if (cand_i[tq] == i) { cand_v[tq] = __builtin_huge_valf(); cand_i[tq] = 0x7fffffff; }
                else { cand_v[kBlock + tq] = __builtin_huge_valf(); cand_i[kBlock + tq] = 0x7fffffff; }
            }
        }
        __syncthreads();
    }

    // the seeds' filter scores: one wave per seed (seed_score_kernel's loop)
    {
        const int v = var.cell_var[cell];
        const int groups = var.groups[v];
        const float* vlib = lib + (size_t)v * var.lib_stride;
        const float* vchroma = lib_chroma + (size_t)v * var.chroma_stride;
        const float4* c0 = reinterpret_cast<const float4*>(c);
        const float4* c1 = c0 + ppad / 4;
        const float4* c2 = c1 + ppad / 4;
        const uint32_t* mk = reinterpret_cast<const uint32_t*>(mkb);
        const float4* cc = reinterpret_cast<const float4*>(cell_chroma + (size_t)cell * ppad);
        for (int slot = wave; slot < M; slot += kBlock / kWave) {
            const int l = picked[slot];
            float acc = __builtin_huge_valf();
            if (l >= 0) {
                const float4* b0 = reinterpret_cast<const float4*>(vlib + (size_t)l * 3 * ppad);
                const float4* b1 = b0 + ppad / 4;
                const float4* b2 = b1 + ppad / 4;
                const float4* bc = reinterpret_cast<const float4*>(vchroma + (size_t)l * ppad);
                acc = 0.0f;
                for (int j = 0; j < groups; ++j) {
                    const int q = lane + kWave * j;
                    const uint32_t m = mk[q];
                    if (m == 0) continue;
                    const float4 p0 = c0[q], p1 = c1[q], p2 = c2[q];
                    const float4 q0 = b0[q], q1 = b1[q], q2 = b2[q];
                    float4 pc{0.f, 0.f, 0.f, 0.f}, qc{0.f, 0.f, 0.f, 0.f};
                    if constexpr (TYPE == CIEDE2000) { pc = cc[q]; qc = bc[q]; }
                    const float w0 = pair_seed<TYPE>(PxF{p0.x, p1.x, p2.x}, PxF{q0.x, q1.x, q2.x}, pc.x, qc.x, lut);
                    const float w1 = pair_seed<TYPE>(PxF{p0.y, p1.y, p2.y}, PxF{q0.y, q1.y, q2.y}, pc.y, qc.y, lut);
                    const float w2 = pair_seed<TYPE>(PxF{p0.z, p1.z, p2.z}, PxF{q0.z, q1.z, q2.z}, pc.z, qc.z, lut);
                    const float w3 = pair_seed<TYPE>(PxF{p0.w, p1.w, p2.w}, PxF{q0.w, q1.w, q2.w}, pc.w, qc.w, lut);
                    acc += ((m & 0x000000ffu) ? w0 : 0.0f) + ((m & 0x0000ff00u) ? w1 : 0.0f)
                         + ((m & 0x00ff0000u) ? w2 : 0.0f) + ((m & 0xff000000u) ? w3 : 0.0f);
                }
                acc = wave_sum_f(acc);
            }
            if (lane == 0) {
                seed_score[(size_t)cell * M + slot] = acc;
                picked_score[slot] = acc;
                if (l >= 0 && scores) scores[(size_t)cell * L + l] = acc;
            }
        }
    }
    __syncthreads();
    // the limit: K-th lowest seed score + the band (prune_limit_kernel), by wave 0
    if (wave == 0) {
        float vv[4];
        for (int k = 0; k < 4; ++k) vv[k] = (lane + kWave * k < M) ? picked_score[lane + kWave * k] : __builtin_huge_valf();
        float kth = __builtin_huge_valf();
        for (int r = 0; r < K; ++r) {
            const float m = fminf(fminf(vv[0], vv[1]), fminf(vv[2], vv[3]));
            float w = m;
            for (int off = 32; off > 0; off >>= 1) w = fminf(w, __shfl_xor(w, off, kWave));
            kth = w;
            const unsigned long long holders = __ballot(m == w);
            if (holders == 0) break;
            if (lane == __ffsll((long long)holders) - 1) {
                if (vv[0] == w) vv[0] = __builtin_huge_valf();
                else if (vv[1] == w) vv[1] = __builtin_huge_valf();
                else if (vv[2] == w) vv[2] = __builtin_huge_valf();
                else vv[3] = __builtin_huge_valf();
            }
        }
        if (lane == 0) {
            const float err = n_active * px_abs + fabsf(kth) * rel;
            limit[cell] = (kth + 2.0f * err) * 1.0001f + 1e-3f;
        }
    }
}

// Pruning limit of one cell from its seeds' filter scores: the K-th lowest seed score
// is an upper bound on the cell's K-th lowest filter score v_K, so nothing scoring
// above  it + 2*err  can be a contender (contenders_kernel keeps filter <= v_K + 2 err).
// One wave per cell, M <= 256 seeds.
__global__ void __launch_bounds__(kWave)
prune_limit_kernel(const float* __restrict__ seed_score, const uint8_t* __restrict__ cmask,
                   int M, int K, int ppad, float px_abs, float rel, float* __restrict__ limit)
{
    const int lane = threadIdx.x;
    const float* sc = seed_score + (size_t)blockIdx.x * M;
    int act = 0;
    const uint32_t* mk = reinterpret_cast<const uint32_t*>(cmask + (size_t)blockIdx.x * ppad);
    for (int q = lane; q < ppad / 4; q += kWave) act += __popc(mk[q] & 0x01010101u);
    for (int off = 32; off > 0; off >>= 1) act += __shfl_xor(act, off, kWave);

    float v[4];                                   // this lane's seeds (M <= 256)
    for (int k = 0; k < 4; ++k) v[k] = (lane + kWave * k < M) ? sc[lane + kWave * k] : __builtin_huge_valf();
    float kth = __builtin_huge_valf();
    for (int r = 0; r < K; ++r) {                 // K rounds: pop the minimum
        float m = fminf(fminf(v[0], v[1]), fminf(v[2], v[3]));
        float w = m;
        for (int off = 32; off > 0; off >>= 1) w = fminf(w, __shfl_xor(w, off, kWave));
        kth = w;
        // remove ONE instance of the minimum: the lowest lane holding it
        const unsigned long long holders = __ballot(m == w);
        if (holders == 0) break;                  // all +inf
        if (lane == __ffsll((long long)holders) - 1) {
            if (v[0] == w) v[0] = __builtin_huge_valf();
            else if (v[1] == w) v[1] = __builtin_huge_valf();
            else if (v[2] == w) v[2] = __builtin_huge_valf();
            else v[3] = __builtin_huge_valf();
        }
    }
    if (lane == 0) {
        const float err = (float)act * px_abs + fabsf(kth) * rel;
        // the seed sums and the filter's running sums are fp32 sums in different orders:
        // a further relative 1e-4 keeps the comparison on the safe side
        limit[blockIdx.x] = (kth + 2.0f * err) * 1.0001f + 1e-3f;
    }
}

// Replaces the fp32 term of every logged discontinuity pair by the fp64
// reference-form value: scores[cell][cand] += (float)(exact - fast).  `fast` must be, bit for bit, the term the
// filter kernel added: prep != 0 when that kernel read the interleaved copies (prepared operands, interleave_kernel).
__global__ void __launch_bounds__(kBlock)
edge_fix_kernel(const float* __restrict__ cells, const float* __restrict__ cell_chroma,
                const float* __restrict__ lib, const float* __restrict__ lib_chroma,
                float* __restrict__ scores, int L, int ppad,
                const EdgeEntry* __restrict__ edges, const unsigned int* __restrict__ edge_count,
                unsigned int edge_cap, const HueEntry* __restrict__ hue_lut, Variants var, int prep)
{
    __shared__ HueEntry lut[kHueLutEntries];
    load_hue_lut(lut, hue_lut);
    const unsigned int n = min(*edge_count, edge_cap);
    for (unsigned int i = blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock) {
        const EdgeEntry en = edges[i];
        const float* c = cells + (size_t)en.cell * 3 * ppad + en.pixel;
        const int v = var.cell_var[en.cell];
        const float* b = lib + (size_t)v * var.lib_stride + (size_t)en.cand * 3 * ppad + en.pixel;
        const PxF p{c[0], c[ppad], c[2 * (size_t)ppad]}, q{b[0], b[ppad], b[2 * (size_t)ppad]};
        bool edge;
        const float cp = cell_chroma[(size_t)en.cell * ppad + en.pixel];
        const float cq = lib_chroma[(size_t)v * var.chroma_stride + (size_t)en.cand * ppad + en.pixel];
        const float fast = prep ? ciede2000_fast<true>(PxF{fmaf(0.5f, p.c0, -25.0f), p.c1, p.c2}, PxF{fmaf(0.5f, q.c0, -25.0f), q.c1, q.c2},
                                                       __fmul_rn(kChromaPrep, cp), __fmul_rn(kChromaPrep, cq), lut, &edge)
                                : ciede2000_fast(p, q, cp, cq, lut, &edge);
        const double exact = ciede2000(Px{p.c0, p.c1, p.c2}, Px{q.c0, q.c1, q.c2});
        atomicAdd(&scores[(size_t)en.cell * L + en.cand], (float)(exact - (double)fast));
    }
}

// The audit.  A candidate that leaves the filter without a full score -- dropped by the bound's stage 0 or 1, by the packed-fp16
// pre-tier's walk or by the fp32 walk's exit test: +inf in its score row -- is seen by neither the pre-tier's canary (survivors only)
// nor the engine's band guard (fully scored candidates only).  So one in `var.audit` of all (cell, candidate) slots -- a hash: the
// same ones every run -- that hold +inf is walked once more here, by the fp32 form WITHOUT any bound: only its own partial sums,
// every term >= 0, can end the walk.  CIEDE2000 pairs on the opposite-hue discontinuity take the fp64 form on the spot (the
// candidates the bound drops first are the ones of opposite hue).  A walk that completes CLEARLY below the limit -- by 1e-3 of it:
// ten times the fp32 form's own error on a cell's sum; a bound within rounding of the limit must not void launches -- means the
// drop was wrong, and the launch is voided like a tripped canary (a sticky bit in the counter block's void word: the engine
// redoes the batch without the filter tiers).  A kernel of its own, after the filter: inside filter_kernel the same check cost
// every candidate (spills: config D +0.75 % with the audit switched off), here only the audited ones.
// TYPE = CIEDE2000: the interleaved copies (`cells` = [B][ppad/256][4][256], `lib` likewise per variant; `cells_f0` unused);
// otherwise (RGB, CIE76: filter_lds_kernel, whose bound carries two constants of its own -- 2e-2 of rounding slack per block sum and
// the factor 0.9999): the planar cell with channel 0 from the NaN-marked plane `cells_f0`, the planar library.
// A workgroup takes 256 slots at a time -- a hash test and a score read per thread -- and then walks the audited ones TOGETHER, wave w
// the groups w, w + 4, ..: a small job's audit is as long as its longest walk (a 128-px cell: 64 groups), which four waves with a
// shared running sum cut to a quarter (config F: 35 -> 12 us, 3.4 % -> 1.1 % of its step).
template <int TYPE>
__global__ void __launch_bounds__(kBlock)
audit_kernel(const float* __restrict__ cells, const float* __restrict__ cells_f0, const float* __restrict__ lib,
             const float* __restrict__ scores, int n_cells, int L, int ppad, const float* __restrict__ prune_above,
             const HueEntry* __restrict__ hue_lut, Variants var, unsigned int* __restrict__ edge_count)
{
    __shared__ HueEntry lut[TYPE == CIEDE2000 ? kHueLutEntries : 1];
    __shared__ int todo_list[kBlock];
    __shared__ int n_todo;
    __shared__ float part[2][kBlock / kWave];
    if constexpr (TYPE == CIEDE2000) load_hue_lut(lut, hue_lut);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const long long total = (long long)n_cells * L;
    const long long n_chunks = (total + kBlock - 1) / kBlock;
    const int pq = ppad / 4;
    for (long long chunk = blockIdx.x; chunk < n_chunks; chunk += gridDim.x) {       // (workgroup-uniform)
        if (threadIdx.x == 0) n_todo = 0;
        __syncthreads();
        const long long slot = chunk * kBlock + threadIdx.x;
        if (slot < total) {
            const int cell = (int)(slot / L), l = (int)(slot - (long long)cell * L);
            if (audit_pick(var.audit, cell, l) && scores[slot] == __builtin_huge_valf()) todo_list[atomicAdd(&n_todo, 1)] = (int)threadIdx.x;
        }
        __syncthreads();
        const int n_here = n_todo;
        for (int t = 0; t < n_here; ++t) {
            const long long s = chunk * kBlock + todo_list[t];
            const int c = (int)(s / L), cand = (int)(s - (long long)c * L);
            const float limit = prune_above[c];
            const int v = var.cell_var[c];
            const int groups = var.groups[v];
            const float4* cb = reinterpret_cast<const float4*>(cells + (size_t)c * (TYPE == CIEDE2000 ? 4 : 3) * ppad);
            const float4* c0 = TYPE == CIEDE2000 ? cb : reinterpret_cast<const float4*>(cells_f0 + (size_t)c * ppad);
            const float4* lb = reinterpret_cast<const float4*>(lib + (size_t)v * (TYPE == CIEDE2000 ? var.chroma_stride * 4 : var.lib_stride) +
                                                               (size_t)cand * (TYPE == CIEDE2000 ? 4 : 3) * ppad);
            float acc = 0.0f;
            bool over = false;
            for (int base = 0, round = 0; base < groups; base += kBlock / kWave, ++round) {
                const int j = base + wave;
                if (j < groups) {
                    if constexpr (TYPE == CIEDE2000) {
                        const int at = 4 * kWave * j + lane;
                        const float4 p0 = cb[at], p1 = cb[at + kWave], p2 = cb[at + 2 * kWave], pc = cb[at + 3 * kWave];
                        const float4 q0 = lb[at], q1 = lb[at + kWave], q2 = lb[at + 2 * kWave], qc = lb[at + 3 * kWave];
                        auto term = [&](float a0, float a1, float a2, float ac, float b0, float b1, float b2, float bc) {
                            bool edge;
                            float d = pair_fast<CIEDE2000, true>(PxF{a0, a1, a2}, PxF{b0, b1, b2}, ac, bc, lut, &edge);
                            // (operands are prepared: channel 0 is L / 2 - 25; an inactive pixel is a NaN there and comes out as 0, never as an edge)
                            if (edge && a0 == a0)
                                d = (float)ciede2000(Px{2.0f * (a0 + 25.0f), a1, a2}, Px{2.0f * (b0 + 25.0f), b1, b2});
                            return d;
                        };
                        acc += (term(p0.x, p1.x, p2.x, pc.x, q0.x, q1.x, q2.x, qc.x) + term(p0.y, p1.y, p2.y, pc.y, q0.y, q1.y, q2.y, qc.y))
                             + (term(p0.z, p1.z, p2.z, pc.z, q0.z, q1.z, q2.z, qc.z) + term(p0.w, p1.w, p2.w, pc.w, q0.w, q1.w, q2.w, qc.w));
                    } else {
                        const int q = kWave * j + lane;
                        const float4 p0 = c0[q], p1 = cb[pq + q], p2 = cb[2 * pq + q];
                        const float4 q0 = lb[q], q1 = lb[pq + q], q2 = lb[2 * pq + q];
                        acc += (euclidean_fast(PxF{p0.x, p1.x, p2.x}, PxF{q0.x, q1.x, q2.x}) + euclidean_fast(PxF{p0.y, p1.y, p2.y}, PxF{q0.y, q1.y, q2.y}))
                             + (euclidean_fast(PxF{p0.z, p1.z, p2.z}, PxF{q0.z, q1.z, q2.z}) + euclidean_fast(PxF{p0.w, p1.w, p2.w}, PxF{q0.w, q1.w, q2.w}));
                    }
                }
                // the four waves' running sums, after every round of four groups (two buffers: one barrier per round)
                const float mine = wave_sum_all(acc);
                if (lane == 0) part[round & 1][wave] = mine;
                __syncthreads();
                const float so_far = (part[round & 1][0] + part[round & 1][1]) + (part[round & 1][2] + part[round & 1][3]);
                if (so_far > limit) { over = true; break; }                          // (workgroup-uniform: every thread reads the same four)
                if (base + kBlock / kWave >= groups && threadIdx.x == 0 && so_far <= limit * (1.0f - 1e-3f)) atomicOr(edge_count + kFilterVoidWord, kVoidByAudit);
            }
            (void)over;
            __syncthreads();        // (the sums' buffers are free again before the next walk writes them)
        }
    }
}

template <int TYPE>
__global__ void __launch_bounds__(kBlock)
pair_fast_kernel(const float* __restrict__ first, const float* __restrict__ second,
                 size_t n, float* __restrict__ out, int flag_edges, const HueEntry* __restrict__ hue_lut)
{
    __shared__ HueEntry lut[TYPE == CIEDE2000 ? kHueLutEntries : 1];
    if constexpr (TYPE == CIEDE2000) load_hue_lut(lut, hue_lut);
    for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n;
         i += (size_t)gridDim.x * kBlock) {
        const float* p = first + 3 * i;
        const float* q = second + 3 * i;
        bool edge;
        float v;
#if MM_WITH_PRETIER
        if (TYPE == CIEDE2000 && (flag_edges & 4) != 0) {   // the packed-fp16 pre-tier's form on operands rounded to halfs as interleave_h_kernel stores them
            auto half_operands = [](const float* c, half2_t& l, half2_t& a, half2_t& b, half2_t& ch) {
                l = h2_splat(fmaf(0.5f, c[0], -25.0f)); a = h2_splat(kHalfOperandScale * c[1]); b = h2_splat(kHalfOperandScale * c[2]);
                ch = h2_splat(kHalfChromaScale * __fmul_rn(kChromaPrep, chroma_of(c[1], c[2])));
            };
            half2_t pl, pa, pb, pch, ql, qa, qb, qch;
            half_operands(p, pl, pa, pb, pch); half_operands(q, ql, qa, qb, qch);
            unsigned int seam = 0u;
            v = (float)ciede2000_h2(pl, pa, pb, pch, ql, qa, qb, qch, lut, &seam).x;
            edge = (seam & 0xffffu) != 0u;                  // counted as 0 by the pre-tier
        } else
#endif
        if (TYPE == CIEDE2000 && (flag_edges & 2) != 0)     // the form the dominant kernel runs: operands prepared as interleave_kernel stores them
            v = pair_fast<TYPE, true>(PxF{fmaf(0.5f, p[0], -25.0f), p[1], p[2]}, PxF{fmaf(0.5f, q[0], -25.0f), q[1], q[2]},
                                      __fmul_rn(kChromaPrep, chroma_of(p[1], p[2])), __fmul_rn(kChromaPrep, chroma_of(q[1], q[2])), lut, &edge);
        else
            v = pair_fast<TYPE>(PxF{p[0], p[1], p[2]}, PxF{q[0], q[1], q[2]}, chroma_of(p[1], p[2]), chroma_of(q[1], q[2]), lut, &edge);
        if (edge && (flag_edges & 1)) v = -1.0f;      // diagnostics: show where the fp64 form takes over
        out[i] = v;
    }
}

// Structured sweep of the fp32 CIEDE2000 filter form against the fp64 form, pairs generated on the
// device (no host arrays: 10^10 pairs take about a second).  What backs the error band the engine
// trusts the filter with (engine.hip filter_band): per pair |fp32 - fp64| <= px_abs + rel * fp64.
//   mode 0: hue1 x hue2 (512 x 512 steps of the circle, jittered) x chroma1 x chroma2 (24 x 24 levels from
//           exactly grey over 1e-4 to 150, jittered) x 25 lightness pairs (L1 in {0,25,50,75,100}, dL in
//           {-40,-5,0,5,40}, jittered) = 3.8e9 pairs per seed;
//   mode 1: the same chroma / lightness grid with hue2 = hue1 + 180 deg + d, d = +-(1e-7 .. 3e-2 rad, 64 log
//           steps): both sides of the opposite-hue discontinuity, where the fp32 branch choice matters.
// Pairs the filter hands to the fp64 form (needs_exact) are counted, not measured.
// out: [0] max |err|  [1] max (|err| - rel * v)  [2] max |err| / (px_abs + rel * v)  [3] max fp64 value
//      as bit patterns of non-negative doubles (atomicMax on u64), [4] handed-over pairs, [5] pairs measured.
__device__ __forceinline__ unsigned int sweep_hash(unsigned long long x)
{
    x ^= x >> 33; x *= 0xff51afd7ed558ccdull; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ull; x ^= x >> 33;
    return (unsigned int)x;
}

__global__ void __launch_bounds__(kBlock)
band_sweep_kernel(int mode, unsigned long long seed, unsigned long long n_pairs, float px_abs, float rel,
                  const HueEntry* __restrict__ hue_lut, unsigned long long* __restrict__ out)
{
    __shared__ HueEntry lut[kHueLutEntries];
    load_hue_lut(lut, hue_lut);
    const float chroma_levels[24] = {0.0f, 1e-4f, 1e-2f, 0.3f, 1.0f, 2.0f, 4.0f, 7.0f, 10.0f, 14.0f, 18.0f, 22.0f,
                                     25.0f, 28.0f, 32.0f, 38.0f, 45.0f, 55.0f, 65.0f, 80.0f, 95.0f, 110.0f, 128.0f, 150.0f};
    const float l_levels[5] = {0.0f, 25.0f, 50.0f, 75.0f, 100.0f}, dl_levels[5] = {-40.0f, -5.0f, 0.0f, 5.0f, 40.0f};
    double m_err = 0.0, m_abs = 0.0, m_ratio = 0.0, m_val = 0.0;
    unsigned long long handed = 0, measured = 0;
    for (unsigned long long i = (unsigned long long)blockIdx.x * kBlock + threadIdx.x; i < n_pairs;
         i += (unsigned long long)gridDim.x * kBlock) {
        unsigned long long r = i;
        const int li = (int)(r % 25); r /= 25;
        const int c2i = (int)(r % 24); r /= 24;
        const int c1i = (int)(r % 24); r /= 24;
        const unsigned int u = sweep_hash(i * 0x9e3779b97f4a7c15ull + seed);
        const float j0 = (float)(u & 1023u) * (1.0f / 1024.0f), j1 = (float)((u >> 10) & 1023u) * (1.0f / 1024.0f),
                    j2 = (float)((u >> 20) & 1023u) * (1.0f / 1024.0f);
        double h1, h2;
        if ((mode & 1) == 0) {
            const int h2i = (int)(r % 512); r /= 512;
            const int h1i = (int)(r % 512);
            h1 = 6.283185307179586 * ((double)h1i + j0) / 512.0;
            h2 = 6.283185307179586 * ((double)h2i + j1) / 512.0;
        } else {
            const int di = (int)(r % 128); r /= 128;
            const int h1i = (int)(r % 512);
            h1 = 6.283185307179586 * ((double)h1i + j0) / 512.0;
            const double mag = 1e-7 * exp(((double)(di >> 1) + j1) * (log(3e5) / 64.0));     // 1e-7 .. 3e-2 rad
            h2 = h1 + 3.141592653589793 + ((di & 1) ? mag : -mag);
        }
        const float ch1 = chroma_levels[c1i] * (1.0f + 0.05f * (j2 - 0.5f)), ch2 = chroma_levels[c2i] * (1.0f + 0.05f * (j0 - 0.5f));
        const float l1 = fminf(fmaxf(l_levels[li / 5] + 6.0f * (j1 - 0.5f), 0.0f), 100.0f);
        const float l2 = fminf(fmaxf(l1 + dl_levels[li % 5] + 2.0f * (j2 - 0.5f), 0.0f), 100.0f);
        const PxF p{l1, (float)(ch1 * cos(h1)), (float)(ch1 * sin(h1))}, q{l2, (float)(ch2 * cos(h2)), (float)(ch2 * sin(h2))};
        bool edge;
        float fast;
#if MM_WITH_PRETIER
        if (mode & 2) {
            // the packed-fp16 pre-tier's form on operands rounded to halfs as interleave_h_kernel stores them; pairs it counts as 0
            // (near the seam) are "handed over".  What must hold is the OVER-estimate: fp16 - fp64 <= band (an under-estimate only
            // delays an exit), so the share reported is the signed one; max_err stays the absolute error
            auto prep = [](const PxF& c, half2_t& l, half2_t& a, half2_t& b, half2_t& ch) {
                l = h2_splat(fmaf(0.5f, c.c0, -25.0f)); a = h2_splat(kHalfOperandScale * c.c1); b = h2_splat(kHalfOperandScale * c.c2);
                ch = h2_splat(kHalfChromaScale * __fmul_rn(kChromaPrep, chroma_of(c.c1, c.c2)));
            };
            half2_t pl, pa, pb, pch, ql, qa, qb, qch;
            prep(p, pl, pa, pb, pch); prep(q, ql, qa, qb, qch);
            unsigned int seam = 0u;
            const half2_t v16 = ciede2000_h2(pl, pa, pb, pch, ql, qa, qb, qch, lut, &seam);
            fast = (float)v16.x;
            edge = (seam & 0xffffu) != 0u;
            if (!edge && !(fast == fast && fabsf(fast) < 1e30f)) m_ratio = 1e9;       // NaN / inf outside the seam: must not happen in the box
        } else
#endif
        {
            fast = ciede2000_fast(p, q, chroma_of(p.c1, p.c2), chroma_of(q.c1, q.c2), lut, &edge);
        }
        if (edge) { handed += 1; continue; }
        const double exact = ciede2000(Px{p.c0, p.c1, p.c2}, Px{q.c0, q.c1, q.c2});
        const double err = fabs((double)fast - exact);
        m_err = fmax(m_err, err);
        m_abs = fmax(m_abs, err - (double)rel * exact);
        m_ratio = fmax(m_ratio, ((mode & 2) ? (double)fast - exact : err) / ((double)px_abs + (double)rel * exact));
        m_val = fmax(m_val, exact);
        measured += 1;
    }
    for (int off = 32; off > 0; off >>= 1) {
        m_err = fmax(m_err, __shfl_xor(m_err, off, kWave)); m_abs = fmax(m_abs, __shfl_xor(m_abs, off, kWave));
        m_ratio = fmax(m_ratio, __shfl_xor(m_ratio, off, kWave)); m_val = fmax(m_val, __shfl_xor(m_val, off, kWave));
        handed += __shfl_xor(handed, off, kWave); measured += __shfl_xor(measured, off, kWave);
    }
    if ((threadIdx.x & 63) == 0) {
        // non-negative doubles order like their bit patterns
        atomicMax(out + 0, (unsigned long long)__double_as_longlong(m_err));
        atomicMax(out + 1, (unsigned long long)__double_as_longlong(fmax(m_abs, 0.0)));
        atomicMax(out + 2, (unsigned long long)__double_as_longlong(m_ratio));
        atomicMax(out + 3, (unsigned long long)__double_as_longlong(m_val));
        atomicAdd(out + 4, handed);
        atomicAdd(out + 5, measured);
    }
}

}  // namespace

hipError_t launch_band_sweep(int mode, unsigned long long seed, unsigned long long n_pairs, float px_abs, float rel,
                             const HueEntry* hue_lut, unsigned long long* out6, hipStream_t stream)
{
    if ((mode & 2) && !MM_WITH_PRETIER) return hipErrorNotSupported;       // (the fp16 form is not in this build)
    hipError_t code = hipMemsetAsync(out6, 0, 6 * sizeof(unsigned long long), stream);
    if (code != hipSuccess) return code;
    // modes 2 / 3: the same two sweeps for the packed-fp16 pre-tier's form, against ITS band
    if ((mode & 2) && !(mode & 4)) { px_abs = kPreBandAbs; rel = kPreBandRel; }       // (bit 2: the caller's own band, diagnostics)
    band_sweep_kernel<<<256 * 16, kBlock, 0, stream>>>(mode, seed, n_pairs, px_abs, rel, hue_lut, out6);
    return hipGetLastError();
}

// Launch order of filter_kernel's cells: the persistent workgroups take the items of an XCD's share front to back, and a
// launch ends when the last item does.  Clipped cells (cell_full != 1: a bound on their whole blocks only, or none) make
// items several times as long as the others; left where the raster order puts them -- the last rows and columns of the grid
// -- they are the tail of the launch (config C: the workgroups were resident 87 % of the longest one's lifetime).  So
// each share's cells are partitioned, stably: long ones first.  One workgroup per share, a contiguous run of cells per thread.
__global__ void __launch_bounds__(kBlock)
cell_order_kernel(const uint8_t* __restrict__ cell_full, int n_cells, int shares, int32_t* __restrict__ order)
{
    __shared__ int n_long[kBlock + 1], n_short[kBlock + 1];
    const int sh = blockIdx.x, t = threadIdx.x;
    const int lo = (int)((long long)n_cells * sh / shares), hi = (int)((long long)n_cells * (sh + 1) / shares);
    const int per = (hi - lo + kBlock - 1) / kBlock;
    const int a = min(lo + t * per, hi), b = min(a + per, hi);
    int longs = 0;
    for (int i = a; i < b; ++i) longs += cell_full[i] != 1;
    n_long[t + 1] = longs;
    n_short[t + 1] = (b - a) - longs;
    __syncthreads();
    if (t == 0) {
        n_long[0] = 0;
        n_short[0] = 0;
        for (int k = 1; k <= kBlock; ++k) { n_long[k] += n_long[k - 1]; n_short[k] += n_short[k - 1]; }
    }
    __syncthreads();
    int at_long = lo + n_long[t], at_short = lo + n_long[kBlock] + n_short[t];
    for (int i = a; i < b; ++i) {
        if (cell_full[i] != 1) order[at_long++] = i;
        else order[at_short++] = i;
    }
}

// audit_kernel / audit_euclid_kernel after a filter launch (in.var.audit says how many; a launch of its own so that the engine
// can clock it apart from the filter).  CIEDE2000: over the interleaved copies only (the default layout)
hipError_t launch_filter_audit(int type, const FilterInputs& in, const float* scores, int n_cells, int L, int ppad,
                               const float* prune_above, unsigned int* edge_count, hipStream_t stream)
{
    if (prune_above == nullptr || in.var.audit == 0u) return hipSuccess;
    const long long chunks = ((long long)n_cells * L + kBlock - 1) / kBlock;
    const dim3 grid((unsigned)std::min<long long>(chunks, 4096));
    if (type == CIEDE2000) {
        if (in.lib_il == nullptr || in.cells_il == nullptr) return hipSuccess;
        audit_kernel<CIEDE2000><<<grid, kBlock, 0, stream>>>(in.cells_il, nullptr, in.lib_il, scores, n_cells, L, ppad, prune_above, in.hue_lut, in.var, edge_count);
    } else {
        // an eighth of the rate (one in 8192 at the default; `audit` = 1 still means every drop): the Euclidean filter drops 99 % of its
        // candidates on the bound alone, so an audited walk costs it relatively more, and its bound is the triangle inequality with two
        // rounding constants, not a derivation with a page of them
        Variants var = in.var;
        if (var.audit > 1u) var.audit = std::min(var.audit * 8u, 1u << 20);
        audit_kernel<RGB_EUCLIDEAN><<<grid, kBlock, 0, stream>>>(in.cells, in.cells_f0, in.lib, scores, n_cells, L, ppad, prune_above, in.hue_lut, var, edge_count);
    }
    return hipGetLastError();
}

hipError_t launch_filter(int type, const FilterInputs& in, float* scores, int n_cells, int L, int ppad,
                         EdgeEntry* edges, unsigned int* edge_count, unsigned int edge_cap,
                         const float* prune_above, hipStream_t stream, int* tile_used)
{
    const int groups = std::max(std::max(in.var.groups[0], in.var.groups[1]), std::max(in.var.groups[2], in.var.groups[3]));
    // Candidates per item and cells per XCD sub-group, measured at 32 / 48 / 64 x 32 / 64 / 128: small cells (config E's
    // 48 x 48: 9 groups) gain 7 % of the launch from 64-candidate items (the survivor mask of the bound's stage 1 holds 64)
    // and 128-cell sub-groups; 64 x 64 cells run the same within 1 % either way, but the larger working set costs
    // 40 % more fabric traffic (303 GB instead of 217 GB per launch), so they keep 32 / 64.
    const bool small_cells = groups < 12;
    // ... rounded down to whole rounds of stage 1: a pass judges 64 / groups candidates, four waves share a tile
    // (48 x 48 cells: 7 per pass -> 56-candidate items, two passes per wave, instead of 64 = 2.3)
    const int round_of = 4 * (kWave / std::max(groups, 1));
    const int small_tile = round_of > 0 && round_of <= 64 ? 64 / round_of * round_of : 64;
    // (cells above 32 groups -- 96 .. 128 px, 150 .. 256 KB interleaved each -- press on the fabric, 0.95 of its peak on config G: what helps them is
    // FEWER cells in flight per XCD, i.e. 16-candidate items and sub-groups of two or three cells, so that the cells stay in the L2 while the
    // library streams through it: config G 15.15 -> 14.48 ms per step on one box, 15.31 -> 14.97 on another, tools/experiments/archive/r5_bigcells.sh)
    const int cand_per_block = in.tile_candidates > 0 ? in.tile_candidates : (small_cells ? small_tile : (groups > 32 && type == CIEDE2000 ? 16 : 32)), cell_group = 512;      // (16: measured on CIEDE2000 only)
    // persistent workgroups of filter_kernel: what the device holds at once (LDS-limited) -- and not one more: a second workgroup
    // per slot ("so that a finishing workgroup's slot refills at once", rounds 1 and 2) starts when the launch is nearly over, pays
    // its start (hue table, prologue, eight counter probes) and finds nothing; measured at twice the slots against once: the
    // reference harness's CIEDE2000 line (config H) 0.488 -> 0.310 ms per launch, config C 2.23 -> 2.02, config D 40.2 -> 39.9
    // ... and fewer than that for a small launch: workgroups start one after the other (~0.1 us each: 0.24 ms for the 2560 of
    // config H, whose launch took 0.49 ms), so with N items of ~20 us the launch is shortest near sqrt(N x 20 / 0.1) workgroups
    // (config H, 5120 items: 0.312 ms at 1280 workgroups, 0.240 at 640, 0.267 at 320)
    auto few = [](long long items, int slots) -> unsigned {
        const long long by_start_cost = std::max<long long>(64, (long long)std::sqrt(100.0 * (double)items));
        return (unsigned)std::min<long long>(std::min<long long>(items, slots), by_start_cost);
    };
    auto persistent_grid = [&](const void* kern, long long items, int threads, size_t dyn_lds) -> unsigned {
        // the occupancy query costs a fraction of a millisecond: asked once per (kernel, LDS size, device) of the PROCESS, not
        // per launch -- and not per thread either: the device group (group.hip) drives every engine from short-lived threads
        struct Known { const void* kern; size_t lds; int dev, slots; };
        static Known known[64];
        static int n_known = 0;
        static std::mutex known_lock;
        int dev = 0;
        (void)hipGetDevice(&dev);
        {
            std::lock_guard<std::mutex> hold(known_lock);
            for (int i = 0; i < n_known; ++i)
                if (known[i].kern == kern && known[i].lds == dyn_lds && known[i].dev == dev)
                    return few(items, known[i].slots);
        }
        int cus = 256, per_cu = 4;
        (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kern, threads, dyn_lds) != hipSuccess || per_cu < 1) per_cu = 4;
        (void)hipGetLastError();
        {
            std::lock_guard<std::mutex> hold(known_lock);
            if (n_known < 64) known[n_known++] = Known{kern, dyn_lds, dev, cus * per_cu * MM_GRID_MULT};
        }
        return few(items, cus * per_cu * MM_GRID_MULT);
    };
    // groups visited by descending bound (the rest in plane order): all of them from 12 groups per cell on (cells of 33 .. 64
    // groups rank the 64 lanes of the prelude; config G: 16.2 -> 15.3 ms per launch, with the exit test at every group 14.4)
    const int sort_groups = in.sort_groups < 0 ? (groups > 16 ? 64 : (groups >= 12 ? 16 : 0)) : in.sort_groups;
    // The LDS-resident cell tile (CIEDE2000, early exit on): ten-wave workgroups, the cell's colour planes staged once
    // per item.  Needs one block per lane in the bound's prelude (<= 16 groups = cells up to 64 x 64) and the three
    // planes beside the hue table, twice per CU.
    if (type == CIEDE2000 && in.use_lds_form && prune_above != nullptr && groups <= kWave / 4 && (size_t)ppad * 12 <= 56 * 1024) {
        constexpr int kTileThreads = 640;
        const int cpb = in.tile_candidates > 0 ? in.tile_candidates : 64;
        if (tile_used) *tile_used = cpb;
        const long long tiles = (L + cpb - 1) / cpb, items = (long long)n_cells * tiles;
        if (items <= 0 || items > 0x7fffffffLL) return hipErrorInvalidValue;
        const int shares = (in.xcd_aware && n_cells >= 8 * 64) ? 8 : 1;
        const int sub = shares > 1 ? (in.xcd_cells > 0 ? in.xcd_cells : 64) : 512;
        hipError_t code = in.counters_cleared ? hipSuccess : hipMemsetAsync(edge_count, 0, kCounterWords * sizeof(unsigned int), stream);
        if (code != hipSuccess) return code;
        auto kern = in.lib_il && in.cells_il ? filter_kernel<CIEDE2000, kTileThreads, true, true> : filter_kernel<CIEDE2000, kTileThreads, true, false>;
        const bool il = in.lib_il && in.cells_il;
        const size_t dyn = (size_t)ppad * 12 + in.lds_pad_bytes;
        if ((code = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn)) != hipSuccess)
            return code;
        kern<<<dim3(persistent_grid(reinterpret_cast<const void*>(kern), items, kTileThreads, dyn)), kTileThreads, dyn, stream>>>(
            in.cells, il ? in.cells_il : in.cells_f0, in.cell_chroma, il ? in.lib_il : in.lib, in.lib_chroma, scores, n_cells, L, ppad, cpb, sub,
            edges, edge_count, edge_cap, prune_above, 1, in.hue_lut, in.var, shares, in.seeds, in.seeds ? in.n_seeds : 0, sort_groups, 1);
        if ((code = hipGetLastError()) != hipSuccess) return code;
        edge_fix_kernel<<<1024, kBlock, 0, stream>>>(in.cells, in.cell_chroma, in.lib, in.lib_chroma, scores, L, ppad,
                                                     edges, edge_count, edge_cap, in.hue_lut, in.var, il ? 1 : 0);
        return hipGetLastError();
    }
    const long long n_tiles = (L + cand_per_block - 1) / cand_per_block;
    const long long blocks = (long long)n_cells * n_tiles;
    if (blocks <= 0 || blocks > 0x7fffffffLL) return hipErrorInvalidValue;
    if (tile_used) *tile_used = cand_per_block;
    // one share of the cells per XCD (8 on MI355X); small launches keep one share so no XCD idles on an empty one
    const int xcd_shares = (in.xcd_aware && n_cells >= 8 * 64) ? 8 : 1;
    // cells per XCD sub-group: the cells an XCD works on at the same time should sit in its 4 MB L2 beside the candidate tiles --
    // 64 cells of 64 x 64 (64 KB each, interleaved), 128 of the smaller ones; cells of 17 .. 32 groups by their size: 2 MB worth of them
    // (80 x 80: 20; flat from 2 to 32 cells, round 5); cells above 32 groups 512 KB worth (128 x 128 = 256 KB a cell: 2; config G, 1024 such
    // cells x 1000 images, per launch: 14.3 ms at 64, 13.4 at 16, 12.7 at 8, 12.5 at 4 in round 4; round 5, with 16-candidate items: 12.1 at 8,
    // 11.9 at 2 and 4)
    const int big_cell_group = groups > 32 ? std::max(2, (int)((size_t)(512 << 10) / ((size_t)ppad * 16)))
                                           : std::max(4, (int)((size_t)(2 << 20) / ((size_t)ppad * 16)));
    const int item_group = xcd_shares > 1 ? (in.xcd_cells > 0 ? in.xcd_cells : (small_cells ? 128 : (groups > 16 ? std::min(64, big_cell_group) : 64))) : cell_group;
    // the exit test after every group wherever the bound reaches (cells up to 64 groups = 128 x 128; config G, 1024 such cells x 1000
    // images: 15.4 ms per launch at every 4th group, 14.4 at every group); larger cells test the plain partial sum 16 times
    const int check_every = in.check_every > 0 ? in.check_every : (groups > kWave ? groups / 16 : 1);
    // items per global atomic.  Measured (round 3, one box): taking several items at a time is SLOWER -- config C 3.53 /
    // 3.64 / 3.70 / 3.91 ms per step at 1 / 2 / 4 / 8, config E's shard 31.8 / 33.3 / 34.0 at 1 / 4 / 8, config D 46.6 / 47.7 /
    // 49.1 at 1 / 2 / 4: the atomic's round trip is hidden behind the other workgroups of the CU, while a workgroup that
    // holds a run of neighbouring cells keeps them from the other workgroups of its XCD (tail imbalance, less sharing of
    // the candidate tile in L2).  One item per atomic stays the default; the option remains for experiments.
    const int item_chunk = in.item_chunk > 0 ? std::min(in.item_chunk, 64) : 1;
    hipError_t code = in.counters_cleared ? hipSuccess
                                          : hipMemsetAsync(edge_count, 0, kCounterWords * sizeof(unsigned int), stream);  // + work counters, item counters
    if (code != hipSuccess) return code;
    // LDS-resident form when the cell (4 planes + mask) fits beside the hue table
    // (Euclid: + the cell's block sums and weights, + the survivor list of a tile of up to 2048 candidates)
    const size_t lds_bytes = type == CIEDE2000 ? (size_t)kHueLutEntries * sizeof(HueEntry) + (size_t)ppad * 16
                                               : (size_t)ppad * 12 + (size_t)4 * (ppad / 64) * 4 + 2048 * 4;
    // RGB / CIE76 cells too large for LDS (from 65 x 65 on; 128 x 128 is the reference GUI's default) with the bound available: the same
    // kernel, the cell read from L2
    const bool cell_global = type != CIEDE2000 && in.use_lds_form && lds_bytes > 156 * 1024 &&
                             (size_t)4 * (ppad / 64) * 4 + 2048 * 4 <= 150 * 1024 &&      // (the block sums of a cell of ~ 750 x 750)
                             in.lib_stats_t != nullptr && prune_above != nullptr;
    if (cell_global || (in.use_lds_form && lds_bytes <= 156 * 1024)) {
        // Candidates per item: a workgroup stages the cell once per item, so the longer the item the less staging and
        // the fewer barriers -- config B (1024 cells x 1000 images): 0.451 / 0.365 / 0.328 / 0.300 ms at 64 / 128 / 256 /
        // 1000 -- as long as the launch still has a few items per workgroup slot of the device (2 x 256 CUs).
        int cpb = in.tile_candidates;
        if (cpb <= 0) {
            const long long want = ((long long)L * n_cells / 2048 + 15) / 16 * 16;
            cpb = (int)std::min<long long>(std::max<long long>(want, 64), std::max(L, 64));
        }
        // with stage 1 (one candidate per lane) a tile should give all 1024 threads a candidate: whole multiples of 1024 up to L
        if (in.tile_candidates <= 0 && in.lib_stats_t != nullptr && prune_above != nullptr && L >= 1024) cpb = std::max(cpb, 1024);
        if (in.tile_candidates <= 0 && in.lib_stats_t != nullptr && prune_above != nullptr && L < 1024) cpb = std::max(cpb, (L + 63) / 64 * 64);
        cpb = std::min(cpb, 2048);              // (the survivor list of a tile)
        if (tile_used) *tile_used = cpb;
        const long long tiles = (L + cpb - 1) / cpb;
        const long long nblk = (long long)n_cells * tiles;
        if (type == CIEDE2000) {
            auto kern = filter_lds_kernel<CIEDE2000>;
            if ((code = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                            (int)lds_bytes)) != hipSuccess) return code;
            kern<<<dim3((unsigned)std::min<long long>(nblk, MM_LDS_GRID_CAP)), kLdsBlock, lds_bytes, stream>>>(
                in.cells, in.cells_f0, in.cell_chroma, in.lib, in.lib_chroma, scores, n_cells, L, ppad, cpb, cell_group,
                edges, edge_count, edge_cap, prune_above, in.hue_lut, in.var, nullptr);
            if ((code = hipGetLastError()) != hipSuccess) return code;
            edge_fix_kernel<<<1024, kBlock, 0, stream>>>(in.cells, in.cell_chroma, in.lib, in.lib_chroma, scores, L, ppad,
                                                         edges, edge_count, edge_cap, in.hue_lut, in.var, 0);
        } else {
            auto kern = cell_global ? filter_lds_kernel<RGB_EUCLIDEAN, true> : filter_lds_kernel<RGB_EUCLIDEAN, false>;
            const size_t lds_used = cell_global ? (size_t)4 * (ppad / 64) * 4 + 2048 * 4 : lds_bytes;
            if ((code = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                            (int)lds_used)) != hipSuccess) return code;
            // (at most 8192 workgroups stride over the items: 16 rounds of what the device holds at once)
            kern<<<dim3((unsigned)std::min<long long>(nblk, MM_LDS_GRID_CAP)), kLdsBlock, lds_used, stream>>>(
                in.cells, in.cells_f0, in.cell_chroma, in.lib, in.lib_chroma, scores, n_cells, L, ppad, cpb, cell_group,
                edges, edge_count, edge_cap, prune_above, in.hue_lut, in.var, in.lib_stats_t);
        }
        return hipGetLastError();
    }
    const bool il = in.lib_il && in.cells_il;
    if (type == CIEDE2000) {
        const bool big = groups > 16 && il;
        auto kern = big ? filter_kernel<CIEDE2000, kBlock, false, true, true>
                        : (il ? filter_kernel<CIEDE2000, kBlock, false, true> : filter_kernel<CIEDE2000, kBlock, false, false>);
#if MM_WITH_PRETIER
        const bool pre = il && in.lib_h != nullptr && in.cells_h != nullptr && prune_above != nullptr;
        if (pre) kern = big ? filter_kernel<CIEDE2000, kBlock, false, true, true, true> : filter_kernel<CIEDE2000, kBlock, false, true, false, true>;
#else
        constexpr bool pre = false;
#endif
        Variants var = in.var;
        var.cell_order = nullptr;
        if (in.cell_order != nullptr && xcd_shares > 1 && prune_above != nullptr && var.cell_full != nullptr) {
            cell_order_kernel<<<xcd_shares, kBlock, 0, stream>>>(var.cell_full, n_cells, xcd_shares, in.cell_order);
            if ((code = hipGetLastError()) != hipSuccess) return code;
            var.cell_order = in.cell_order;
        }
        if (in.lds_pad_bytes > 0 &&
            (code = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)in.lds_pad_bytes)) != hipSuccess) return code;
        // (the grid first: its one-time occupancy query is host work that a clocked launch must not have between its two events)
        const unsigned grid_n = persistent_grid(reinterpret_cast<const void*>(kern), blocks, kBlock, in.lds_pad_bytes);
        if (in.clock_begin != nullptr && (code = hipEventRecord(in.clock_begin, stream)) != hipSuccess) return code;
        kern<<<dim3(grid_n), kBlock, in.lds_pad_bytes, stream>>>(
            pre ? reinterpret_cast<const float*>(in.cells_h) : (il ? nullptr : in.cells), il ? in.cells_il : in.cells_f0, il ? nullptr : in.cell_chroma,
            il ? in.lib_il : in.lib, pre ? reinterpret_cast<const float*>(in.lib_h) : (il ? nullptr : in.lib_chroma), scores, n_cells, L, ppad, cand_per_block,
            item_group, edges, edge_count, edge_cap, prune_above, check_every, in.hue_lut, var, xcd_shares,
            in.seeds, in.seeds ? in.n_seeds : 0, sort_groups, item_chunk);
        if ((code = hipGetLastError()) != hipSuccess) return code;
        edge_fix_kernel<<<1024, kBlock, 0, stream>>>(in.cells, in.cell_chroma, in.lib, in.lib_chroma, scores, L, ppad,
                                                     edges, edge_count, edge_cap, in.hue_lut, in.var, il ? 1 : 0);
        if ((code = hipGetLastError()) != hipSuccess) return code;
        if (in.clock_end != nullptr && (code = hipEventRecord(in.clock_end, stream)) != hipSuccess) return code;
    } else {
        filter_kernel<RGB_EUCLIDEAN, kBlock, false, false><<<dim3(persistent_grid(reinterpret_cast<const void*>(filter_kernel<RGB_EUCLIDEAN, kBlock, false, false>), blocks, kBlock, 0)), kBlock, 0, stream>>>(
            in.cells, in.cells_f0, in.cell_chroma, in.lib, in.lib_chroma, scores, n_cells, L, ppad, cand_per_block,
            item_group, edges, edge_count, edge_cap, prune_above, check_every, in.hue_lut, in.var, xcd_shares,
            in.seeds, in.seeds ? in.n_seeds : 0, sort_groups, item_chunk);
    }
    return hipGetLastError();
}

hipError_t launch_interleave(const float* planar, const float* chroma, const float* plane0, float* out,
                             size_t n_images, int ppad, hipStream_t stream)
{
    const size_t total = n_images * (size_t)(ppad / 4);
    const unsigned blocks = (unsigned)std::min<size_t>((total + kBlock - 1) / kBlock, 65536);
    if (blocks == 0) return hipSuccess;
    interleave_kernel<<<blocks, kBlock, 0, stream>>>(planar, chroma, plane0, out, n_images, ppad);
    return hipGetLastError();
}

hipError_t launch_interleave_h(const float* il, void* out_halfs, size_t n_images, int ppad, hipStream_t stream)
{
#if MM_WITH_PRETIER
    const size_t n_vec = n_images * (size_t)ppad;                  // 4 planes x ppad floats = ppad float4
    const unsigned blocks = (unsigned)std::min<size_t>((n_vec + kBlock - 1) / kBlock, 65536);
    if (blocks == 0) return hipSuccess;
    interleave_h_kernel<<<blocks, kBlock, 0, stream>>>(reinterpret_cast<const float4*>(il), reinterpret_cast<uint2*>(out_halfs), n_vec);
    return hipGetLastError();
#else
    (void)il; (void)out_halfs; (void)n_images; (void)ppad; (void)stream;
    return hipErrorNotSupported;
#endif
}

hipError_t launch_group_sums(const float* planar, const uint8_t* mask, size_t mask_stride, float* stats,
                             size_t block_stride, size_t n_images, int ppad, int slots_per_image, hipStream_t stream)
{
    const int n_groups = ppad / 256, n_blocks = slots_per_image;
    const size_t waves = n_images * (size_t)n_groups;
    if (waves == 0) return hipSuccess;
    group_sums_kernel<<<(unsigned)((waves + 3) / 4), kBlock, 0, stream>>>(planar, mask, mask_stride, stats, block_stride,
                                                                            n_images, n_groups, n_blocks, ppad);
    return hipGetLastError();
}

hipError_t launch_image_stats(float* stats, size_t block_stride, int slots_per_image, int image_slot, int active_blocks,
                              size_t n_images, hipStream_t stream, const uint8_t* var_of_image, int4 blocks_of_var)
{
    if (n_images == 0) return hipSuccess;
    image_stats_kernel<<<(unsigned)((n_images + kBlock / kWave - 1) / (kBlock / kWave)), kBlock, 0, stream>>>(stats, block_stride, slots_per_image, image_slot,
                                                                                             active_blocks, n_images, var_of_image, blocks_of_var);
    return hipGetLastError();
}

hipError_t launch_transpose_stats(const float* stats, size_t block_stride, int slots_per_image, int n_blocks, float* out,
                                  size_t n_images, hipStream_t stream)
{
    const size_t total = (size_t)3 * n_blocks * n_images;
    if (total == 0) return hipSuccess;
    transpose_stats_kernel<<<(unsigned)std::min<size_t>((total + kBlock - 1) / kBlock, 65536), kBlock, 0, stream>>>(
        stats, block_stride, slots_per_image, n_blocks, out, n_images);
    return hipGetLastError();
}

hipError_t launch_chroma_planes(const float* planar, float* chroma, size_t n_images, int ppad, hipStream_t stream)
{
    const size_t total = n_images * (size_t)(ppad / 4);
    const unsigned blocks = (unsigned)std::min<size_t>((total + kBlock - 1) / kBlock, 65536);
    if (blocks == 0) return hipSuccess;
    chroma_planes_kernel<<<blocks, kBlock, 0, stream>>>(planar, chroma, n_images, ppad);
    return hipGetLastError();
}

hipError_t launch_image_means(const float* planar, float* means, int n_images, int P, int ppad, hipStream_t stream)
{
    image_means_kernel<<<n_images, kBlock, 0, stream>>>(planar, means, P, ppad);
    return hipGetLastError();
}

hipError_t launch_prune_limits(int type, const FilterInputs& in, const float* lib_means, int n_cells, int L,
                               int ppad, int M, int K, float px_abs, float rel, int32_t* seeds,
                               float* seed_score, float* limit, float* scores, hipStream_t stream)
{
    if (M > 256 || M < K) return hipErrorInvalidValue;
    if (in.fused_seeds) {
        if (type == CIEDE2000)
            seed_fused_kernel<CIEDE2000><<<n_cells, kBlock, 0, stream>>>(in.cells, in.cmask, in.cell_chroma, in.lib, in.lib_chroma, lib_means,
                                                                         L, ppad, M, K, px_abs, rel, seeds, seed_score, limit, scores, in.hue_lut, in.var);
        else
            seed_fused_kernel<RGB_EUCLIDEAN><<<n_cells, kBlock, 0, stream>>>(in.cells, in.cmask, in.cell_chroma, in.lib, in.lib_chroma, lib_means,
                                                                             L, ppad, M, K, px_abs, rel, seeds, seed_score, limit, scores, in.hue_lut, in.var);
        return hipGetLastError();
    }
    const size_t lds = 0;
    const long long waves = (long long)n_cells * M;
    const unsigned score_blocks = (unsigned)std::min<long long>((waves + 3) / 4, MM_SEED_GRID_CAP);   // grid-stride: the hue table once per workgroup
    if (type == CIEDE2000) {
        seed_kernel<CIEDE2000><<<n_cells, kBlock, lds, stream>>>(in.cells, in.cmask, lib_means, L, ppad, M, seeds, in.hue_lut);
        seed_score_kernel<CIEDE2000><<<score_blocks, kBlock, 0, stream>>>(
            in.cells, in.cmask, in.cell_chroma, in.lib, in.lib_chroma, seeds, seed_score, n_cells, M, ppad, in.hue_lut, in.var, scores, L);
    } else {
        seed_kernel<RGB_EUCLIDEAN><<<n_cells, kBlock, lds, stream>>>(in.cells, in.cmask, lib_means, L, ppad, M, seeds, in.hue_lut);
        seed_score_kernel<RGB_EUCLIDEAN><<<score_blocks, kBlock, 0, stream>>>(
            in.cells, in.cmask, in.cell_chroma, in.lib, in.lib_chroma, seeds, seed_score, n_cells, M, ppad, in.hue_lut, in.var, scores, L);
    }
    hipError_t code = hipGetLastError();
    if (code != hipSuccess) return code;
    prune_limit_kernel<<<n_cells, kWave, 0, stream>>>(seed_score, in.cmask, M, K, ppad, px_abs, rel, limit);
    return hipGetLastError();
}

hipError_t launch_pair_fast(int type, const float* first, const float* second, size_t n,
                            float* out, int flag_edges, const HueEntry* hue_lut, hipStream_t stream)
{
    if ((flag_edges & 4) && !MM_WITH_PRETIER) return hipErrorNotSupported;  // (the fp16 form is not in this build)
    const int blocks = (int)((n + kBlock - 1) / kBlock < 8192 ? (n + kBlock - 1) / kBlock : 8192);
    if (type == CIEDE2000)
        pair_fast_kernel<CIEDE2000><<<blocks, kBlock, 0, stream>>>(first, second, n, out, flag_edges, hue_lut);
    else
        pair_fast_kernel<RGB_EUCLIDEAN><<<blocks, kBlock, 0, stream>>>(first, second, n, out, flag_edges, hue_lut);
    return hipGetLastError();
}

}  // namespace mm
