// launch.hpp -- host-side launchers of the kernels that live in filter.hip (a
// separate translation unit because it is built with fp contraction enabled).
#pragma once

#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

// The packed-fp16 pre-tier of the CIEDE2000 filter (colour_diff_fp16.hpp, filter_pretier.hpp): 0 = not compiled in (the default
// build: no PRE instantiations of filter_kernel, no half copies, the options "pretier" / "pretier_band" accepted and ignored,
// mm_debug_has_pretier() == 0); 1 = round 5's code (tools/build_variant.sh NAME -DMM_WITH_PRETIER=1 with MM_ENGINE_FLAGS the same).
#ifndef MM_WITH_PRETIER
#define MM_WITH_PRETIER 0
#endif

namespace mm {

// A pixel pair the fp32 CIEDE2000 form must not judge (opposite-hue discontinuity).
struct EdgeEntry { int cell, cand, pixel; };

// Mask variants.  The 4 flip variants of the cell mask are de-duplicated into n <= 4
// distinct ones; for each, library and cell planes are stored in a PERMUTED pixel order
// with the variant's active pixels first (row-major among themselves), so the scoring
// kernels walk only groups[v] = ceil(active / 256) float4-groups per lane instead of
// ppad / 256.  cell_var[c] = variant of cell c's mask in the current batch.
struct Variants {
    const uint8_t* cell_var;    // [B]
    size_t lib_stride;          // floats between the planar libraries of two variants (L*3*ppad)
    size_t chroma_stride;       // floats between their chroma planes (L*ppad)
    int groups[4];
    // lower bound of the filter's early exit (filter.hip): seven statistics per image and
    // 64-position block over the mask's active positions (group_sums_kernel), as seven arrays
    // Per image n_blocks = ppad / 64 + 4 slots: the blocks, then (slot ppad / 64) the same statistic over the WHOLE image
    // (image_stats_kernel; stage 0 of the bound), then padding that keeps every image's slots 16-byte aligned.
    const float* lib_stats;     // 7 x [V][L][n_blocks], lib_stats_stride floats apart
    const float* cell_stats;    // 8 x [B][n_blocks], cell_stats_stride floats apart (7: block wholly active in the cell, 1 / 0)
    size_t lib_stats_stride, cell_stats_stride;
    int n_blocks;               // slots per image: ppad / 64 + 4
    const uint8_t* cell_full;   // [B] 1 = the cell's active set is its whole mask; 2 = clipped, with a per-block flag (cell statistic 7)
                                // saying which blocks still are whole; 0 = clipped, no flags (separate gather kernels): no bound
    const int32_t* cell_order;  // [B] filter_kernel's launch order of the cells (launch_filter fills it: within each XCD's share the
                                // clipped cells, whose weaker bound makes their items several times as long, come first); null: 0, 1, 2 ..
    int two_stage;              // bit 0: the bound is evaluated several candidates per wave pass ahead of the pixel loops (filter_kernel
                                // stage 1); bit 1: the whole-image bound ahead of that (stage 0; needs the image slots filled)
    unsigned int audit;         // filter_kernel (CIEDE2000): N = 2^k > 0 -- one in N of the candidates that leave without a full score (dropped by the
                                // bound, by the packed-fp16 pre-tier's walk or by the fp32 walk) is walked once more by the fp32 form without the
                                // bound, and a launch in which such a candidate turns out to lie clearly below the limit is voided; 0: none
    float pre_abs, pre_rel;     // the pre-tier's band per pixel pair, absolute and relative part (kPreBandAbs / kPreBandRel: the worst-case bound of
                                // tools/filter_error_bound.py; option "pretier_band" 1: round 4's observed band; diagnostics scale both, a negative
                                // band drops true contenders -- what the audit is for)
    float bound_scale;          // 1; diagnostics: the early exit's lower bound times this (above 1: a bound that drops true contenders)
};

// Device inputs of the fp32 filter kernels.
struct FilterInputs {
    Variants var;
    const float* cells;         // f32 planar [B][3][ppad]
    const uint8_t* cmask;       // u8 [B][ppad]
    const float* cells_f0;      // f32 [B][ppad]  channel 0 with NaN at inactive pixels (what the filter kernels read)
    const float* cell_chroma;   // f32 [B][ppad]  sqrt(a^2+b^2) per cell pixel (CIEDE2000 only)
    const float* lib;           // f32 planar [L][3][ppad]
    const float* lib_chroma;    // f32 [L][ppad]
    const float2* hue_lut;      // [kHueLutSize + 2] {T, -2 sin(2 d_theta)}  (colour_diff_fast.hpp)
    int use_lds_form;           // 1: cell tile staged in LDS by 1024-thread workgroups when it fits
    int tile_candidates;        // candidates per workgroup of filter_kernel (0 = default)
    size_t lds_pad_bytes;       // diagnostics: unused dynamic LDS per filter workgroup (lowers the occupancy)
    int xcd_aware;              // 1: persistent workgroups take the cells of their own XCD's share first (L2 affinity)
    int xcd_cells;              // cells per sub-group of an XCD's share (0 = default: 64, or 128 for cells below 12 groups)
    int sort_groups;            // n > 0: a candidate's n largest-bound groups are visited first, by descending bound, the rest in
                                // plane order (cells up to 64 x 64); 0: plane order
    const float* lib_il;        // interleaved copies for filter_kernel<IL> (launch_interleave): [V][L][ppad/256][4][256] and
    const float* cells_il;      // [B][ppad/256][4][256] (plane 0 = cells_f0); null: the planar arrays are read
    const void* lib_h;          // the interleaved copies as halfs (launch_interleave_h), for the packed-fp16 pre-tier of
    const void* cells_h;        // filter_kernel<.., PRE>; null: no pre-tier
    const float* lib_stats_t;   // Euclid LDS kernel's stage 1: the library's three channel sums per block, candidate-fastest
                                // [variant][3][ppad / 64][L] (launch_transpose_stats); null: one wave per candidate from the start
    int check_every;            // groups between two exit tests of a candidate (0 = 1 up to 16 groups per cell, groups / 16 above)
    int item_chunk;             // work items a persistent workgroup takes per global atomic (0 = by launch size)
    int counters_cleared;       // 1: the counter block was zeroed by an earlier kernel of the stream (no memset here)
    int fused_seeds;            // 1: seeds, their scores and the limit in one launch (launch_prune_limits)
    int32_t* cell_order;        // workspace [B] for Variants::cell_order (null: the cells are taken in their given order)
    const int32_t* seeds;       // [B][n_seeds] candidates launch_prune_limits scored already (their scores are in the
    int n_seeds;                // score rows): filter_kernel skips them.  null / 0: none.  n_seeds <= 256.
    hipEvent_t clock_begin, clock_end;   // non-null: recorded directly around filter_kernel (+ its seam patch) of the general CIEDE2000 path, after every
                                         // host-side one-time query of the launch (the engine's pre-tier trial compares two such intervals)
};

// Words of the counter block launch_filter clears and filter_kernel fills (u32 indices into edge_count):
// [0] discontinuity pairs logged; [1] the VOID word: sticky bits (atomicOr -- a counter pushed past a capacity could wrap) saying
// that the launch's scores must not be trusted: kVoidByCanary (the pre-tier's canary), kVoidByAudit (a dropped candidate the audit
// found below the limit); [2,3] [4,5] [6,7] u64: pixel groups visited, groups of a full visit,
// candidates dropped by the block bound alone; [32], [64] .. [256] item counters of the persistent workgroups (a cache line each).
constexpr int kFilterCounterWords = 288;
constexpr int kFilterVoidWord = 1;
// The packed-fp16 pre-tier's band (filter.hip): worst-case bound with a tenth of room | twice the worst observed pair of round 4.
constexpr float kPreBandAbs = 0.45f, kPreBandRel = 3.8e-2f, kPreBandAbsObserved = 0.29f, kPreBandRelObserved = 8e-3f;
constexpr unsigned int kVoidByCanary = 1u, kVoidByAudit = 2u;

// Tier-1 filter scores f32 [n_cells][L] (see filter.hip).  Pairs on the discontinuity
// are logged to `edges` (capacity edge_cap, count in *edge_count, which may exceed
// the capacity: the caller must then not trust the scores) and patched with the
// fp64 form by a follow-up kernel on the same stream.  prune_above (may be null):
// per-cell limit above which a candidate is abandoned and scored +inf.
// tile_used (may be null): candidates per work item the launch ran with.
hipError_t launch_filter(int type, const FilterInputs& in, float* scores, int n_cells, int L, int ppad,
                         EdgeEntry* edges, unsigned int* edge_count, unsigned int edge_cap,
                         const float* prune_above, hipStream_t stream, int* tile_used = nullptr);
// The audit of a launch_filter (filter.hip audit_kernel, audit_euclid_kernel): one in in.var.audit of the (cell, candidate) slots
// that the launch left at +inf is walked again by the fp32 form without the bound; a walk that completes clearly below the
// cell's limit sets kVoidByAudit in edge_count[kFilterVoidWord] (the caller must then not trust the scores).  A no-op without limits, with
// audit = 0, and for CIEDE2000 without interleaved copies.
hipError_t launch_filter_audit(int type, const FilterInputs& in, const float* scores, int n_cells, int L, int ppad,
                               const float* prune_above, unsigned int* edge_count, hipStream_t stream);

// stats array k (block_stride floats apart), [img][ppad / 64] over plane positions [64 b, 64 b + 64):
// masked sum of channel 0, max |channel 0 - 50|, sums of channels 1 and 2, max chroma, hue range.
// (slots_per_image: floats between two images' statistics, >= ppad / 64)
hipError_t launch_group_sums(const float* planar, const uint8_t* mask, size_t mask_stride, float* stats,
                             size_t block_stride, size_t n_images, int ppad, int slots_per_image, hipStream_t stream);

// Whole-image statistics from the block statistics, written into each image's slot ppad / 64 of the same arrays: sums added
// up (in fp64), maxima taken, the hue arcs of the first `active_blocks` blocks joined into one arc around the direction of
// the image's summed (a, b) (the whole circle if a block's arc is open or reaches further than 170 deg from that direction).
// var_of_image (may be null): the mask variant of every image (cells of a batch differ in theirs); the blocks joined are then
// blocks_of_var[variant] instead of `active_blocks` -- four masks need not hold the same number of active pixels.
hipError_t launch_image_stats(float* stats, size_t block_stride, int slots_per_image, int image_slot, int active_blocks,
                              size_t n_images, hipStream_t stream, const uint8_t* var_of_image = nullptr,
                              int4 blocks_of_var = make_int4(0, 0, 0, 0));

// Candidate-fastest copy of the three channel sums (statistics 0, 2, 3) of n images: out [3][n_blocks][n_images].
hipError_t launch_transpose_stats(const float* stats, size_t block_stride, int slots_per_image, int n_blocks, float* out,
                                  size_t n_images, hipStream_t stream);

// The interleaved copy of n images as halfs ([image][group][plane][256], a and b / 4): what the packed-fp16 pre-tier reads.
hipError_t launch_interleave_h(const float* il, void* out_halfs, size_t n_images, int ppad, hipStream_t stream);
// Interleaved copy of n planar images (+ chroma, + optional replacement of channel 0): [n][ppad/256][4][256].
hipError_t launch_interleave(const float* planar, const float* chroma, const float* plane0, float* out,
                             size_t n_images, int ppad, hipStream_t stream);

// Chroma planes of n planar images: [n][3][ppad] -> [n][ppad].
hipError_t launch_chroma_planes(const float* planar, float* chroma, size_t n_images, int ppad, hipStream_t stream);

// Mean colour per library image, f32 [n][3] (a proxy for choosing seeds; never a score).
hipError_t launch_image_means(const float* planar, float* means, int n_images, int P, int ppad,
                              hipStream_t stream);

// Per-cell pruning limit for launch_filter: M seeds per cell picked by mean-colour
// proximity are scored in full; the K-th lowest of them bounds the cell's K-th lowest
// score from above, so a candidate whose running sum passes limit[cell] cannot be a
// contender.  seeds [n][M] i32, seed_score [n][M] f32, limit [n] f32 are workspaces /
// outputs.  scores (the filter's f32 [n][L] rows): each seed's score is also left at its place there, so that
// launch_filter, given the seed lists (FilterInputs::seeds), need not score those candidates again.
// Requires K <= M <= 256; any library size (the M seeds are the closest of the two best of each of 256 stripes of the
// library: exact up to 512 images, a deterministic near-top-M beyond).
hipError_t launch_prune_limits(int type, const FilterInputs& in, const float* lib_means, int n_cells, int L,
                               int ppad, int M, int K, float px_abs, float rel, int32_t* seeds,
                               float* seed_score, float* limit, float* scores, hipStream_t stream);

// fp32 filter form of the per-pair difference (accuracy harness).  flag_edges bit 0: write -1 for pairs the filter hands
// to the fp64 form; bit 1: the form with prepared operands (L / 2 - 25, chroma / 50), as filter_kernel runs it.
hipError_t launch_pair_fast(int type, const float* first, const float* second, size_t n,
                            float* out, int flag_edges, const float2* hue_lut, hipStream_t stream);

// Structured device-side sweep of the fp32 CIEDE2000 filter form against the fp64 form (filter.hip
// band_sweep_kernel): out6 = {max |err|, max (|err| - rel v), max |err| / (px_abs + rel v), max v} as
// double bit patterns, pairs handed to the fp64 form, pairs measured.
hipError_t launch_band_sweep(int mode, unsigned long long seed, unsigned long long n_pairs, float px_abs, float rel,
                             const float2* hue_lut, unsigned long long* out6, hipStream_t stream);

}  // namespace mm
