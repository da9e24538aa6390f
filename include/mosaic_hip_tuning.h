/*
 * mosaic_hip_tuning.h -- tuning options and diagnostics of libmosaic_hip.so.
 *
 * NOT part of what a HIPPhotomosaicGenerator binds (that is mosaic_hip.h: lifetime, staging, generate, build,
 * group, errors, timing and three switches).  Everything here either trades TIME for time -- the grid mm_generate
 * returns is the same whatever these are set to (tests/test_gpu_fuzz.py::test_tuning_options_do_not_change_the_result)
 * -- or is a diagnostic the test suite and tools/ use to measure the engine against itself.  Options marked [!]
 * deliberately BREAK a guarantee (a wrong band, a wrong bound): they exist so that tests can show the audit, the
 * band guard and the canary catch exactly that, and must never be set by a host.
 * The measured defaults and what each knob bought are in DESIGN.md section 4 and HISTORY.md.
 */
#ifndef MOSAIC_HIP_TUNING_H
#define MOSAIC_HIP_TUNING_H

#include "mosaic_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Option names of mm_set_option beyond "filter" / "prune" / "max_batch" (name, default, one line each).
 *
 * -- the repeat-penalty resolve ------------------------------------------------------------------------------------
 *   sweep_resolve      1     the raster-order dependency as parallel sweeps to its fix point
 *   max_sweeps         48    sweeps in total before the sequential chain takes over
 *   front_resolve      1     where a first set of sweeps does not settle (photographs): the exact wavefront in ONE launch, built for latency,
 *                            and at once on the next generate over the same library; 0 = further sets of sweeps, then round 1's row pipeline
 *   wave_resolve       1     the chain's single-wave form (0: workgroup per cell)
 *   row_pipeline       1     the wavefront forms where the grid fits LDS (0: the single-wave chain)
 *   spec_k             4     length of mm_generate's speculative short lists; 0 = always mm_candidates_per_cell
 *   spec_auto          1     the short lists' length by measurement: a generate that deepened > 2 % of its cells lets the next ones over the same job try
 *                            twice and four times the length (<= 16); the length whose whole generate was fastest is kept; 0 pins spec_k
 *   warm_rounds        1     the resolve after a round that extended only a few lists starts its sweeps from the previous round's grid; 0 = from a cleared one
 *   deepen_mid         16    a cell whose short list did not verify its pick is first extended to n entries (>= twice the short length),
 *                            to the full counting length only if that did not verify either; 0 = straight to the full length
 * -- submission ------------------------------------------------------------------------------------------------------
 *   fast_path          1     a whole generate as ONE submission and one result block (one host wait)
 *   fused              1     gather + chroma + interleave + block statistics in one launch; contenders + re-score + re-rank in one
 *   fused_seeds        2048  batches of up to n cells pick seeds, score them and set the limit in one launch
 *   phase_timing       -1    events between the phases (mm_timing's *_ms): 1 always, 0 never, -1 for calls that do not notice ~35 us
 *   spin_wait_us       1000  the host polls the stream this long before it blocks on it
 *   tiny               1     detail cells of <= 64 pixels take the one-wave-per-cell fp64 path
 * -- the filter kernel's work items ------------------------------------------------------------------------------------
 *   tile               0     candidates per work item; 0 = by cell size (and by item_auto)
 *   item_auto          1     item size follows the share of candidates the bound dropped in the engine's previous launch over the library
 *   item_chunk         0     work items a persistent workgroup takes per global atomic (0 = 1)
 *   xcd_aware          1     per-XCD item order (hwreg XCC_ID)
 *   xcd_cells          0     cells per XCD sub-group; 0 = by cell size
 *   cell_order         1     each XCD share's clipped cells (long items) first
 *   interleave         1     [group][plane] image copies with prepared operands
 *   lds_filter         -1    LDS-resident cell tile: -1 = RGB / CIE76 only, 1 = CIEDE2000 too, 0 = never
 *   lds_pad_kb         0     extra dynamic LDS per workgroup (occupancy experiments)
 * -- the early exit ------------------------------------------------------------------------------------------------------
 *   seeds              -1    proxy-chosen seeds beyond the list length; -1 = 14 (CIEDE2000) / 2 (RGB, CIE76)
 *   prune_min_kpairs   -1    below n thousand pixel pairs per call seeds and early exit are skipped; -1 = 1e5 (RGB) / 2e4 (CIEDE2000)
 *   stage0             1     one whole-image bound per candidate ahead of the block bound
 *   two_level          1     the block bound's stage 1: several candidates per wave pass
 *   euclid_stage1      1     RGB / CIE76: the block bound one candidate per lane before any wave is spent
 *   sort_groups        -1    how many of a candidate's groups are visited by descending bound; -1 = by cell size
 *   check_every        0     pixel groups between two exit tests; 0 = by cell size
 * -- the guards (rates; results identical, what changes is how often a check runs) ---------------------------------------
 *   band_guard         1     every n-th cell's best EXCLUDED candidate rides through the fp64 re-score (mm_timing.band_alarms); 0 = off
 *   audit              1024  one in n candidates that left the filter without a full score is walked again without the bound; 0 = off
 * -- the packed-fp16 pre-tier (compiled in only with -DMM_WITH_PRETIER=1; mm_debug_has_pretier) ------------------------------
 *   pretier            -1    -1 = measured per (library, cell size) where eligible, 0 never, 1 always; ignored when not compiled in
 *   pretier_band       0     0 = the worst-case band of the fp16 form, 1 = round 4's observed band (a statistical guarantee)
 * -- diagnostics ---------------------------------------------------------------------------------------------------------
 *   probe_prepared     0     what mm_filter_diff_batch / mm_colour_diff_batch measure: 1 prepared operands, 2 the fp16 form, 3 the algebraic fp64 form
 *   band_pct           100   [!] the fp32 filter's trusted band in per cent
 *   dbg_bound_pct      100   [!] the early exit's lower bound in per cent (above 100 drops contenders: what the audit catches)
 *   dbg_pre_band_pct   100   [!] the pre-tier's band in per cent (negative drops contenders)
 *   dbg_pre_abs_1e4 / dbg_pre_rel_1e6   0   the band mm_debug_filter_band_sweep modes 2 / 3 measure against; 0 = the pre-tier's own
 */

/* 1 if the library was built with the packed-fp16 pre-tier (-DMM_WITH_PRETIER=1; the default build leaves it out:
 * on its certified band it loses on every benchmarked configuration). */
int mm_debug_has_pretier(void);

/* Diagnostics: the fp32 FILTER form of the same difference (tier 1 of the two-tier
 * scoring).  It never decides a result; this entry exists so that its
 * error against mm_colour_diff_batch can be measured (tests/test_gpu_filter.py). */
mm_status mm_filter_diff_batch(mm_engine *e, int diff_type,
                               const float *first, const float *second,
                               size_t n, float *out);

/* Diagnostics: what backs the error band the engine trusts the fp32 filter with.  A structured
 * sweep generated and reduced on the device -- mode 0: hue x hue x chroma x chroma x lightness
 * (3.8e9 distinct grid points per seed, jittered by `seed`); mode 1: the same chroma / lightness
 * grid on both sides of the opposite-hue discontinuity (hue difference 180 deg +- 1e-7 .. 3e-2 rad).
 * out: max |fp32 - fp64|, max (|err| - 6e-5 v), max |err| / (band px_abs + 6e-5 v) -- the share of
 * the trusted band the worst pair uses, must stay below 1 --, the largest fp64 value met, pairs
 * handed to the fp64 form, pairs measured.  Modes 2 and 3: the same two sweeps for the filter's packed-fp16 pre-tier
 * (operands rounded to halfs as the engine stores them) against ITS band; the share is then the signed OVER-estimate
 * (fp16 - fp64) / band -- an under-estimate only delays an early exit --, "handed over" counts the pairs the form counts as 0
 * near the opposite-hue seam, and a non-finite value off the seam reports a share of 1e9. */
mm_status mm_debug_filter_band_sweep(mm_engine *e, int mode, uint64_t seed, uint64_t n_pairs, double out[6]);

/* Diagnostics: {pixel groups visited by the filter kernel, groups a full visit takes,
 * candidates dropped outright by the bound} of the last filter launch (always counted; the
 * sums over a generate are in mm_timing). */
mm_status mm_debug_visit_counters(mm_engine *e, unsigned long long out[3]);

/* Calibration: GB/s a plain coalesced float4 read stream over `bytes` of device memory reaches
 * (best of `repeats` launches); bench.py quotes the roofline fraction against it as well as
 * against the 8 TB/s specification. */
mm_status mm_debug_stream_read(mm_engine *e, size_t bytes, int repeats, double *gb_per_s);

#ifdef __cplusplus
}
#endif
#endif /* MOSAIC_HIP_TUNING_H */
