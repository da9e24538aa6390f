// engine.hip -- host orchestration + C ABI of the best-fit engine (include/mosaic_hip.h).
//
// Replaces the device half of CUDAPhotomosaicGenerator::generateBestFits
// (src/Photomosaic/CUDA/CUDAPhotomosaicGenerator.cpp:40-368).  What is NOT
// reproduced from it, on purpose: per-generate cudaMalloc of everything
// (:372-510), one H2D per cell (:180-185), an L*P*8-byte memset per cell
// (:213), >= 2 launches + 1 D2D per (cell, library image) (:234-250), the
// serial <<<1,1>>> repeat / lowest kernels.  Here the library, masks and target
// are staged once, cells are gathered on the device, one launch scores a whole
// batch of cells against the whole library, and the raster-order dependency of
// the repeat penalty is confined to a small resolve pass over K candidates.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <limits>
#include <mutex>
#include <new>
#include <string>
#include <vector>

#include "kernels.hpp"
#include "colour_diff_fast.hpp"
#include "launch.hpp"

namespace {

thread_local std::string g_create_error;

struct DevBuf {
    void* p = nullptr;
    size_t bytes = 0;
};

}  // namespace

struct mm_engine {
    int device = 0;
    hipStream_t stream = nullptr;
    hipEvent_t ev[15] = {};      // [12], [13]: around the filter launch of a pre-tier trial (below); [14]: after the filter's audit
    std::string err;

    // staged inputs
    DevBuf lib;      int n_lib = 0, lib_size = 0;
    DevBuf masks;    int mask_size = 0;
    DevBuf target;   int rows = 0, cols = 0;

    // per-call workspaces (grown on demand, kept)
    DevBuf recs, cells, cmask, scores, cand_i, cand_s, grid, best, counter, scratch;
    // two-tier scoring workspaces
    DevBuf cont_i, cont_f, cont_s, flags, edges, edge_count, dense_recs, dense_i, sentinel;
    DevBuf lib_means, seeds, seed_score, limits, row_slot, dense_rows_buf, row_index;
    DevBuf lib_chroma, cell_chroma, hue_lut, lib_il, cells_il;
    // pretier = -1: WHETHER the pre-tier pays is a property of the content and of its band -- on the synthetic benchmark candidates leave the
    // pixel loop far above the limit and the pre-tier's cheaper arithmetic wins with round 4's observed band (config D -6.3 %), with the
    // worst-case band of round 5 nowhere (config D +9 %, config G +5 %: there -1 means off); on photographs the block bound is tight,
    // many candidates sit within the fp16 band of the limit, walk all their groups in fp16 and then again in fp32 (the photo fixture
    // enlarged to 2752 / 6144 cells: +6 % / +4 %, tools/experiments/archive/r4_photo_scale.py).  So an eligible job MEASURES it: after a warm-up
    // launch over a (library, cell size, batch size, target), one filter launch runs without the pre-tier and one with it, both between
    // two events, and the faster form is kept from there on.  As with the item size and the list length only the TIME of a call
    // depends on the engine's history, never its result.
    struct PreAuto { int lib = 0, size = 0, cells = 0, have = 0, pending = -1, choice = -1; bool warm = false; float ms[2] = {0.f, 0.f}; } pre_auto;
    DevBuf lib_h, cells_h;       // the interleaved copies as halfs (the filter's packed-fp16 pre-tier)
    bool lib_h_ready = false;    // lib_h holds the current library (prepare_planes builds it only where the pre-tier may run)
    float dbg_pre_abs = 0.0f, dbg_pre_rel = 0.0f;      // diagnostics: the band mm_debug_filter_band_sweep modes 2 / 3 measure against (0: the pre-tier's own)
    int spin_wait_us = 1000;     // mm_set_option("spin_wait_us", n): the host polls the stream for up to n microseconds before it blocks on it (0: blocks at once)
    int audit = 1024;            // mm_set_option("audit", N): one in N (a power of two) of the CIEDE2000 candidates that leave the filter without a full score
                                 // (bound, fp16 walk, fp32 walk) is walked again by the fp32 form without the bound; a wrong drop voids the launch; 0: none
    int pre_band_pct = 100;      // diagnostics ("dbg_pre_band_pct"): the pre-tier's band in per cent of the validated one (negative: drops contenders)
    int pretier_band = 0;        // mm_set_option("pretier_band", 0 / 1): 0 = the worst-case band (tools/filter_error_bound.py: every rounding of the fp16
                                 // form at its worst, a tenth of room); 1 = round 4's observed band (twice the worst pair of 1.1e10 swept and climbed
                                 // pairs: a third of the worst-case bound -- a STATISTICAL guarantee, which is why it is not the default)
    int bound_pct = 100;         // diagnostics ("dbg_bound_pct"): the early exit's lower bound in per cent (above 100: drops contenders)
    int pretier = -1;            // mm_set_option("pretier", 0 / 1): the packed-fp16 pre-tier ahead of the fp32 filter form never / always;
                                 // -1 = off on the worst-case band; with the observed band ("pretier_band" 1) measured (PreAuto above) for jobs
                                 // with something to gain -- cells above 32 pixel groups (up to 128 x 128, the reference GUI's default) and calls of >= 1e10 pixel
                                 // pairs (config D 45.7 -> 42.8, E's shard 30.1 -> 29.4, 80-px cells 11.7 -> 11.0) --, off for smaller
                                 // ones (config C 3.21 -> 3.36).  Results are identical either way
    DevBuf row_map, spec_recs, deep, insufficient, sweep_counts;
    DevBuf stage_u8, srgb_lut, shrink_tmp, covered, cells_f0;
    DevBuf main_bgr;   int main_rows = 0, main_cols = 0;      // mm_set_main_image_bgr: the u8 main image (entropy test)
    DevBuf entropy_out, exceeded_out;
    int mosaic_rows = 0, mosaic_cols = 0;       // the mosaic kept between mm_build_photomosaic_step calls
    int cell_scale = 1;          // mm_set_cell_scale / mm_set_cell_resize: normal-cell pixels per detail pixel and
                                 // axis when that is a whole number, 0 = any ratio (area table in `area_tab`)
    int cell_normal = 0;         // normal cell size the area table was built for
    int area_detail = 0;         // detail cell size the area table was built (and laid out) for
    DevBuf area_tab;             // [detail + 1] first | [entries] src | [entries] alpha (kernels.hpp CellResize)
    int area_entries = 0;
    int spec_k = 4;              // mm_set_option("spec_k", n): entries of the speculative short lists (0 = off)
    int spec_auto = 1;           // mm_set_option("spec_auto", 0): the list length no longer follows what the engine's previous generates over the
                                 // same job measured (mm_generate learn_list_length)
    int spec_k_learned = 0, spec_lib = 0, spec_size = 0;     // ... what it learned, and for which library (count, cell size)
    int spec_cells = 0;          // ... and cell count: the lengths are compared by the TIME of whole generates over the same job
    float spec_ms[3] = {0.f, 0.f, 0.f};     // fastest generate seen at the base length, twice and four times it (0: not tried)
    int spec_base_runs = 0;      // generates timed at the base length (the first one over a library runs cold: it gets a second, warm one before the verdict)
    int warm_rounds = 1;         // mm_set_option("warm_rounds", 0): a later round's resolve starts from a cleared grid (and goes to the front form at once where that was learned)
    int deepen_mid = 16;         // mm_set_option("deepen_mid", n): a cell whose short list could not verify its pick is first extended to n entries
                                 // (at least twice the short length) and only then to the full counting length; 0: straight to the full length
    int sweeps_learned = 0, sweeps_lib = 0, sweeps_size = 0; // sweeps the resolve of the previous generate over this library took to settle (its longest
                                                             // round): sizes the first set of sweeps of the next one (resolve_single_submission)
    DevBuf lib_bgr, owner, mosaic;   int n_lib_bgr = 0, lib_bgr_size = 0;
    // mask variants (launch.hpp Variants): library kept as handed over (HWC) and re-planed
    // per distinct mask in that mask's active-pixels-first order whenever either changes
    DevBuf lib_hwc, perm, inv_perm, cell_var, cell_full, cell_order, pos_mask, lib_lsum, cell_lsum;
    std::vector<uint8_t> masks_host;
    bool planes_dirty = true;
    int n_var = 0, var_of_mask[4] = {0, 0, 0, 0}, var_groups[4] = {0, 0, 0, 0};
    int use_row_pipeline = 1;    // mm_set_option("row_pipeline", 0): single serial chain
    int use_front_resolve = 1;   // mm_set_option("front_resolve", 0): resolve_rows_kernel as the wavefront form, and further sets of sweeps instead of it
    bool front_learned = false;  // the sweeps of a generate over (sweeps_lib, sweeps_size) did not settle in their first set: the front form at once
    bool result_grid_valid = false;   // the device result block holds the final grid of the last single-submission resolve (a later round of the
                                 // same generate may start its sweeps from it)
    bool last_front = false;     // the last resolve_single_submission ran the front form (its "sweep" count says nothing about the sweeps' depth)
    int use_sweep_resolve = 1;   // mm_set_option("sweep_resolve", 0): raster-order chain instead of parallel sweeps
    int max_sweeps = 48;         // mm_set_option("max_sweeps", n): sweeps before falling back to the chain
    int use_lds_filter = -1;     // mm_set_option("lds_filter", 0/1); -1 = auto (Euclid only: it is the L2-bound one)
    int extra_seeds = -1;        // mm_set_option("seeds", n): proxy-chosen seeds beyond the list length; -1 = 14 for CIEDE2000, 2 for RGB / CIE76
    int lds_pad_kb = 0;          // mm_set_option("lds_pad_kb", n): diagnostics -- extra LDS per filter workgroup (lowers occupancy)
    int tile_candidates = 0;     // mm_set_option("tile", n): candidates per filter workgroup (0 = default)
    double drop_rate = -1.0;     // share of the candidates the bound dropped in the last CIEDE2000 filter launch over ...
    int drop_rate_lib = 0, drop_rate_size = 0;      // ... this library (count, cell size): picks the item size of the next launch
    int max_batch_cells = 0;     // mm_set_option("max_batch", n): cap on cells per device batch (0 = from memory)
    int use_wave_resolve = 1;    // mm_set_option("wave_resolve", 0): workgroup-per-cell resolve
    int use_filter = 1;          // mm_set_option("filter", 0) forces the fp64-only path
    int use_prune = 1;           // mm_set_option("prune", 0): filter pass without early exit
    int two_level = 1;           // mm_set_option("two_level", 0): the bound is evaluated per candidate by the wave that walks its pixels,
                                 // not four candidates per wave pass ahead of the pixel loops (filter_kernel's two stages)
    int sort_groups = -1;        // mm_set_option("sort_groups", n): the n largest-bound pixel groups of a candidate are visited first,
                                 // in descending order of their bounds, the others in plane order (0: plane order, 16: all sorted);
                                 // -1 = auto: all sorted from 12 groups per cell on (measured: 64 x 64 cells -6 % launch time,
                                 // 48 x 48 cells and the hexagon mask +0..2 %: fewer visits, but a worse L2 hit rate)
    int use_interleave = 1;      // mm_set_option("interleave", 0): the filter kernel reads the planar arrays (eight bases per pair)
    int xcd_aware = 1;           // mm_set_option("xcd_aware", 0): one item order for the whole device
    int xcd_cells = 0;           // mm_set_option("xcd_cells", n): cells per sub-group of an XCD's share (0 = default)
    bool any_clipped = false;        // the cells of the call in flight: does any of them carry a bound short of the whole cell? (check_cells)
    bool long_items_first = true;    // mm_set_option("cell_order", 0 / 1): filter_kernel takes the clipped cells of each XCD's share first
    int probe_prepared = 0;      // mm_set_option("probe_prepared", 0 / 1 / 2): mm_filter_diff_batch measures the prepared-operand form of the CIEDE2000 filter (1) or the packed-fp16 pre-tier's form (2); 3: mm_colour_diff_batch measures the algebraic fp64 form of the parallel sums
    int check_every = 0;         // mm_set_option("check_every", n): groups between two exit tests of a candidate (0 = by cell size)
    int item_chunk = 0;          // mm_set_option("item_chunk", n): work items per global atomic of the filter's persistent workgroups (0 = by launch size)
    size_t batch_per_cell = 0;   // workspace bytes per cell the batch capacity below was computed for
    int batch_capacity = 0;      // cells per device batch the last hipMemGetInfo allowed (0 = not asked yet)
    int band_pct = 100;          // mm_set_option("band_pct", n): diagnostics -- trust only n % of the filter band (tests of the
                                 // guards; a negative value turns the band inside out: true contenders get excluded)
    int band_guard_every = 1;    // mm_set_option("band_guard", n): every n-th cell of a batch (default: EVERY cell) re-scores the best candidate the
                                 // filter band excluded in fp64 (0 = off); an excluded candidate that wins is a band failure
    int euclid_stage1 = 1;       // mm_set_option("euclid_stage1", 0): the RGB / CIE76 filter gives every candidate a wave from the start
    DevBuf lib_stats_t;          // [V][3][ppad / 64][L]: the library's channel sums per block, candidate-fastest (filter_lds_kernel stage 1)
    int stage0 = 1;              // mm_set_option("stage0", 0): no whole-image bound ahead of the block bound (filter_kernel stage 0)
    int var_active[4] = {0, 0, 0, 0}; // active pixels of each mask variant
    int tiny_path = 1;           // mm_set_option("tiny", 0): detail cells of <= 64 pixels go through the general path too
    DevBuf lib_tiny;             // position-major library copy [variant][64][3][L] for tiny_cells_kernel
    int item_auto = 1;           // mm_set_option("item_auto", 0): the filter's item size no longer follows the drop rate of the
                                 // engine's previous CIEDE2000 launch (timing then does not depend on the call history)
    int fused = 1;               // mm_set_option("fused", 0): separate gather / chroma / interleave / statistics and
                                 // contenders / re-score / re-rank launches instead of the two fused kernels
    int fast_path = 1;           // mm_set_option("fast_path", 0): mm_generate asks the host between its phases (step-by-step
                                 // path) instead of enqueuing a whole generate and reading ONE result block back
    DevBuf result;               // [ResultHeader][grid i64][scores f64] of the single-submission path
    void* pinned = nullptr;      // page-locked host mirror of `result`
    size_t pinned_bytes = 0;
    // a score batch enqueued but not yet looked at by the host (single-submission path)
    struct Pending { bool active = false, two_tier = false, prune = false, counted = false; int b = 0, diff_type = 0; unsigned int edge_cap = 0; } pend;
    int64_t syncs = 0;           // stream synchronisations of the current call (mm_timing.host_syncs)
    mm::ResultHeader* fast_hdr = nullptr;   // device header of the single-submission generate in flight (else null)
    int phase_timing = -1;       // mm_set_option("phase_timing", n): events between the phases of a call (mm_timing's *_ms):
                                 // 1 always, 0 never, -1 = only for calls of >= ~2e10 CIEDE2000 pixel pairs or ten times as many
                                 // RGB ones (the nine events cost ~35 us per call)
    bool phase_events = true;    // ... as decided for the current call
    int fused_seed_cells = 2048; // mm_set_option("fused_seeds", n): batches of up to n cells pick seeds, score them and set the
                                 // limit in ONE launch (0 = always the three separate kernels)
    long long prune_min_pairs = -1;   // mm_set_option("prune_min_kpairs", n): below n thousand pixel pairs per batch the
                                      // seeds / early exit are skipped (three launches cost more than they save); -1 = default

    // value_range_kernel's verdict on the staged f32 inputs (bits: 1 non-finite, 2 outside the CIEDE2000 box, 4 outside the
    // Euclidean box the filter band was validated on); u8 inputs are converted here and lie inside by construction
    unsigned int lib_range = 0, target_range = 0;
    DevBuf range_flags;

    mm_timing timing{};
    mm_stage_timing stage_clock{};    // host wall clocks of everything that is NOT the hot path (mm_get_stage_timing): what a first call pays
};

namespace {

void set_err(mm_engine* e, const char* fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (e) e->err = buf; else g_create_error = buf;
}

// The shape of createCUDAErrStr (CUDA/CUDAUtility.h:27-30) + the expected /
// unexpected split of CUDAErrMessageBox (:32-62).
mm_status hip_fail(mm_engine* e, hipError_t code, const char* what, const char* file, int line)
{
    set_err(e, "%s: %s (%d) at %s:%d", what, hipGetErrorName(code), (int)code, file, line);
    (void)hipGetLastError();
    return code == hipErrorOutOfMemory ? MM_OUT_OF_MEMORY : MM_DEVICE_ERROR;
}

#define HIP_TRY(e, call, what)                                                   \
    do {                                                                          \
        hipError_t hip_try_code_ = (call);                                        \
        if (hip_try_code_ != hipSuccess)                                          \
            return hip_fail((e), hip_try_code_, (what), __FILE__, __LINE__);      \
    } while (0)

inline double wall_ms()
{
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

// Adds the host wall time of its scope to one field of the engine's stage clock.
struct StageScope {
    double* field;
    double t0;
    explicit StageScope(double* f) : field(f), t0(wall_ms()) {}
    ~StageScope() { if (field) *field += wall_ms() - t0; }
};

mm_status ensure(mm_engine* e, DevBuf& b, size_t bytes, const char* what)
{
    if (b.bytes >= bytes && b.p) return MM_OK;
    StageScope clock(e ? &e->stage_clock.alloc_ms : nullptr);
    if (e) { e->stage_clock.allocations += 1; e->stage_clock.allocated_bytes += (int64_t)bytes; }
    if (b.p) { (void)hipFree(b.p); b.p = nullptr; b.bytes = 0; }
    if (bytes == 0) bytes = 16;
    HIP_TRY(e, hipMalloc(&b.p, bytes), what);
    b.bytes = bytes;
    return MM_OK;
}

void release(DevBuf& b)
{
    if (b.p) (void)hipFree(b.p);
    b.p = nullptr;
    b.bytes = 0;
}

// Relative gap under which two fp64 totals count as a near-tie and are re-summed in
// the reference's order.  The parallel-order sums differ from the sequential ones
// by <= ~P * 2^-53 (~5e-13 at P = 4096); 1e-9 leaves three orders of margin.
constexpr double kTieBand = 1e-9;

inline int round_up(int v, int m) { return (v + m - 1) / m * m; }
// Slots per image of the block statistics: the ppad / 64 blocks, the whole-image slot, padding to a float4 (launch.hpp Variants).
inline int stat_slots(int ppad) { return ppad / 64 + 4; }


inline bool known_type(int t) { return t == MM_RGB_EUCLIDEAN || t == MM_CIE76 || t == MM_CIEDE2000; }

mm_status check_ready(mm_engine* e)
{
    if (e->n_lib <= 0 || e->mask_size <= 0 || e->rows <= 0) {      // (staged = the logical state; buffers may be adopted ones)
        set_err(e, "library, masks and target must be staged before generating");
        return MM_NOT_READY;
    }
    if (e->lib_size != e->mask_size) {
        set_err(e, "library image size %d != mask size %d (detail cell size must match, "
                   "PhotomosaicGeneratorBase.cpp:262-271)", e->lib_size, e->mask_size);
        return MM_INVALID_ARGUMENT;
    }
    if (e->cell_scale == 0 && e->area_detail != e->lib_size) {
        // the resize table is indexed by detail pixels: a library that changed size since
        // mm_set_cell_resize would read it out of bounds (cell_resize_of)
        set_err(e, "the cell resize table was built for a detail size of %d but the library is %d px now: "
                   "call mm_set_cell_resize again", e->area_detail, e->lib_size);
        return MM_NOT_READY;
    }
    return MM_OK;
}

mm_status check_cells(mm_engine* e, const mm_cell* cells, int n, int grid_w, int grid_h)
{
    const int size = e->lib_size;
    e->any_clipped = false;
    for (int i = 0; i < n; ++i) {
        const mm_cell& c = cells[i];
        e->any_clipped = e->any_clipped || c.row_min != 0 || c.col_min != 0 || c.row_max != size || c.col_max != size;
        const bool ok =
            c.row_min >= 0 && c.row_min < c.row_max && c.row_max <= size &&
            c.col_min >= 0 && c.col_min < c.col_max && c.col_max <= size &&
            c.mask_id >= 0 && c.mask_id < 4 &&
            (e->cell_scale != 1 ||     // the resize zero-fills what lies outside the image
             ((long long)c.y + c.row_min >= 0 && (long long)c.y + c.row_max <= e->rows &&
              (long long)c.x + c.col_min >= 0 && (long long)c.x + c.col_max <= e->cols)) &&
            (e->cell_scale == 1 || (std::abs((long long)c.x) < (1LL << 28) && std::abs((long long)c.y) < (1LL << 28))) &&
            (grid_w <= 0 || (c.grid_x >= 0 && c.grid_x < grid_w && c.grid_y >= 0 && c.grid_y < grid_h));
        if (!ok) {
            set_err(e, "cell %d: bound [%d,%d)x[%d,%d) at (%d,%d), mask %d, grid (%d,%d) does not fit "
                       "a %dx%d cell on a %dx%d image / %dx%d grid", i, c.row_min, c.row_max,
                    c.col_min, c.col_max, c.x, c.y, c.mask_id, c.grid_x, c.grid_y, size, size,
                    e->cols, e->rows, grid_w, grid_h);
            return MM_INVALID_ARGUMENT;
        }
    }
    return MM_OK;
}

// How the kernels make a detail pixel (kernels.hpp CellResize) from the engine's state.
mm::CellResize cell_resize_of(const mm_engine* e)
{
    mm::CellResize r{e->cell_scale, nullptr, nullptr, nullptr};
    if (e->cell_scale == 0) {
        const int* base = (const int*)e->area_tab.p;
        r.first = base;
        r.src = base + (e->lib_size + 1);
        r.alpha = (const float*)(base + (e->lib_size + 1) + e->area_entries);
    }
    return r;
}

// The engine's device made current, and the calling thread's sticky HIP error cleared: the kernel launches below are
// checked with hipGetLastError(), which would otherwise report an error some OTHER library left behind on this thread
// (RCCL probing peer access while it builds a communicator leaves hipErrorInvalidDevice, for one).
hipError_t select_device(mm_engine* e)
{
    const hipError_t code = hipSetDevice(e->device);
    (void)hipGetLastError();
    return code;
}

// The host's wait for the stream.  A blocking hipStreamSynchronize wakes the thread through an interrupt, tens of microseconds after the
// last kernel has ended -- a tenth of a sub-millisecond generate (profiles/r04_small_job_gaps.txt) --, so the first `spin_wait_us`
// microseconds are spent polling instead (one core, bounded: a 43 ms step polls for its first millisecond and then blocks as before).
hipError_t sync_stream(mm_engine* e)
{
    e->syncs += 1;
    if (e->spin_wait_us > 0) {
        const auto t0 = std::chrono::steady_clock::now();
        const auto budget = std::chrono::microseconds(e->spin_wait_us);
        do {
            const hipError_t q = hipStreamQuery(e->stream);
            if (q != hipErrorNotReady) {
                if (q == hipSuccess) (void)hipGetLastError();       // (a hipErrorNotReady of an earlier poll must not look like an error of a later launch)
                return q;
            }
        } while (std::chrono::steady_clock::now() - t0 < budget);
        (void)hipGetLastError();
    }
    return hipStreamSynchronize(e->stream);
}

float elapsed(hipEvent_t a, hipEvent_t b)
{
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, a, b);
    return ms;
}

template <typename F>
mm_status by_type(int diff_type, F&& f)
{
    if (diff_type == MM_CIEDE2000) return f(std::integral_constant<int, mm::CIEDE2000>{});
    return f(std::integral_constant<int, mm::RGB_EUCLIDEAN>{});   // CIE76 == Euclid on Lab
}

// The value range of f32 pixels handed over as they are (mm_set_library / mm_set_target): is the fp32 filter's band
// valid for them?  One pass over the resident copy, four bytes back.
mm_status value_range(mm_engine* e, const float* hwc_dev, size_t n_pixels, unsigned int* flags_out)
{
    mm_status st = ensure(e, e->range_flags, sizeof(unsigned int), "Failed to allocate range flags");
    if (st) return st;
    HIP_TRY(e, hipMemsetAsync(e->range_flags.p, 0, sizeof(unsigned int), e->stream), "Failed to clear range flags");
    const unsigned blocks = (unsigned)std::min<size_t>((n_pixels + mm::kBlock - 1) / mm::kBlock, 4096);
    mm::value_range_kernel<<<std::max(blocks, 1u), mm::kBlock, 0, e->stream>>>(hwc_dev, n_pixels, (unsigned int*)e->range_flags.p);
    HIP_TRY(e, hipGetLastError(), "Error during value range kernel");
    HIP_TRY(e, hipMemcpyAsync(flags_out, e->range_flags.p, sizeof(unsigned int), hipMemcpyDeviceToHost, e->stream), "range flags");
    HIP_TRY(e, sync_stream(e), "Failed to synchronise stream");
    return MM_OK;
}

// Whether the two-tier path may run on what is staged: the filter's band covers the validated box only.
inline bool filter_band_covers(const mm_engine* e, int diff_type)
{
    const unsigned int f = e->lib_range | e->target_range;
    return (f & (diff_type == MM_CIEDE2000 ? 3u : 5u)) == 0u;
}

// De-duplicates the 4 mask variants, builds each distinct variant's pixel permutation
// (active pixels first, in a tile-major walk among themselves, then the inactive ones) and re-planes
// the library in that order: [V][L][3][ppad] + chroma planes [V][L][ppad] + means.
mm_status prepare_planes(mm_engine* e)
{
    if (!e->planes_dirty) return MM_OK;
    // (wall time of this function minus the allocations it makes, which the allocation clock has already)
    struct PlanesClock {
        mm_engine* e; double t0, alloc0;
        ~PlanesClock() { e->stage_clock.planes_ms += (wall_ms() - t0) - (e->stage_clock.alloc_ms - alloc0); }
    } planes_clock{e, wall_ms(), e->stage_clock.alloc_ms};
    const int size = e->lib_size, P = size * size, ppad = round_up(P, 256), L = e->n_lib;
    int n_var = 0;
    int first_mask_of_var[4];
    for (int m = 0; m < 4; ++m) {
        int v = -1;
        for (int u = 0; u < n_var && v < 0; ++u) {
            const uint8_t* a = e->masks_host.data() + (size_t)m * P;
            const uint8_t* b = e->masks_host.data() + (size_t)first_mask_of_var[u] * P;
            bool same = true;
            for (int p = 0; p < P && same; ++p) same = (a[p] != 0) == (b[p] != 0);
            if (same) v = u;
        }
        if (v < 0) { v = n_var; first_mask_of_var[n_var++] = m; }
        e->var_of_mask[m] = v;
    }
    std::vector<int> perm((size_t)n_var * ppad, -1), inv((size_t)n_var * P, 0);
    std::vector<uint8_t> pos_mask((size_t)n_var * ppad, 0);     // 1 on the plane positions holding active pixels
    // Pixel walk: 16 x 16 tiles in raster order, inside a tile its four 8 x 8 quarters, raster inside a
    // quarter.  A 256-position group of the planes is then a compact 16 x 16 patch of the image (for a
    // full mask) and a 64-position block a compact 8 x 8 one, which is what makes the block means of
    // the filter's early-exit bound (filter.hip) tell neighbouring images apart.
    std::vector<int> walk(P);
    {
        std::vector<std::pair<unsigned, int>> keyed(P);
        for (int p = 0; p < P; ++p) {
            const unsigned y = (unsigned)(p / size), x = (unsigned)(p % size);
            const unsigned tiles_x = ((unsigned)size + 15u) / 16u;
            const unsigned key = ((((y / 16u) * tiles_x + x / 16u) * 4u + ((y % 16u) / 8u) * 2u + (x % 16u) / 8u) * 64u)
                               + (y % 8u) * 8u + x % 8u;
            keyed[p] = {key, p};
        }
        std::sort(keyed.begin(), keyed.end());
        for (int p = 0; p < P; ++p) walk[p] = keyed[p].second;
    }
    for (int v = 0; v < n_var; ++v) {
        const uint8_t* mk = e->masks_host.data() + (size_t)first_mask_of_var[v] * P;
        int k = 0;
        for (int p : walk) if (mk[p] != 0) { perm[(size_t)v * ppad + k] = p; inv[(size_t)v * P + p] = k; ++k; }
        for (int a = 0; a < k; ++a) pos_mask[(size_t)v * ppad + a] = 1;
        e->var_groups[v] = (k + 255) / 256;
        e->var_active[v] = k;
        for (int p : walk) if (mk[p] == 0) { perm[(size_t)v * ppad + k] = p; inv[(size_t)v * P + p] = k; ++k; }
    }
    for (int v = n_var; v < 4; ++v) e->var_groups[v] = 0;
    e->n_var = n_var;

    mm_status st;
    if ((st = ensure(e, e->perm, perm.size() * sizeof(int), "Failed to allocate pixel order")) ||
        (st = ensure(e, e->inv_perm, inv.size() * sizeof(int), "Failed to allocate pixel order")) ||
        (st = ensure(e, e->lib, (size_t)n_var * L * 3 * ppad * sizeof(float), "Failed to allocate library planes")) ||
        (st = ensure(e, e->lib_chroma, (size_t)n_var * L * ppad * sizeof(float), "Failed to allocate library chroma planes")) ||
        (st = ensure(e, e->lib_means, (size_t)L * 3 * sizeof(float), "Failed to allocate library means")) ||
        (st = ensure(e, e->pos_mask, pos_mask.size(), "Failed to allocate position masks")) ||
        (st = ensure(e, e->lib_lsum, (size_t)7 * n_var * L * stat_slots(ppad) * sizeof(float), "Failed to allocate block statistics")))
        return st;
    HIP_TRY(e, hipMemcpyAsync(e->pos_mask.p, pos_mask.data(), pos_mask.size(), hipMemcpyHostToDevice, e->stream),
            "Failed to copy position masks to device");
    HIP_TRY(e, hipMemcpyAsync(e->perm.p, perm.data(), perm.size() * sizeof(int), hipMemcpyHostToDevice, e->stream),
            "Failed to copy pixel order to device");
    HIP_TRY(e, hipMemcpyAsync(e->inv_perm.p, inv.data(), inv.size() * sizeof(int), hipMemcpyHostToDevice, e->stream),
            "Failed to copy pixel order to device");
    for (int v = 0; v < n_var; ++v) {
        const int blocks = (int)std::min<size_t>(((size_t)L * ppad + mm::kBlock - 1) / mm::kBlock, 65536);
        mm::hwc_to_planar_kernel<<<blocks, mm::kBlock, 0, e->stream>>>(
            (const float*)e->lib_hwc.p, (float*)e->lib.p + (size_t)v * L * 3 * ppad, L, P, ppad,
            (const int*)e->perm.p + (size_t)v * ppad);
        HIP_TRY(e, hipGetLastError(), "Error during library plane kernel");
        HIP_TRY(e, mm::launch_group_sums((const float*)e->lib.p + (size_t)v * L * 3 * ppad,
                                         (const uint8_t*)e->pos_mask.p + (size_t)v * ppad, 0,
                                         (float*)e->lib_lsum.p + (size_t)v * L * stat_slots(ppad),
                                         (size_t)n_var * L * stat_slots(ppad), (size_t)L, ppad, stat_slots(ppad), e->stream),
                "Error during library group-sum kernel");
    }
    HIP_TRY(e, mm::launch_chroma_planes((const float*)e->lib.p, (float*)e->lib_chroma.p, (size_t)n_var * L, ppad, e->stream),
            "Error during library chroma kernel");
    HIP_TRY(e, mm::launch_image_means((const float*)e->lib.p, (float*)e->lib_means.p, L, P, ppad, e->stream),
            "Error during library mean kernel");
    // the interleaved copy the dominant kernel reads ([group][plane][256]: one base address per image)
    if ((st = ensure(e, e->lib_il, (size_t)n_var * L * 4 * ppad * sizeof(float), "Failed to allocate interleaved library"))) return st;
    HIP_TRY(e, mm::launch_interleave((const float*)e->lib.p, (const float*)e->lib_chroma.p, nullptr, (float*)e->lib_il.p,
                                     (size_t)n_var * L, ppad, e->stream), "Error during library interleave kernel");
    e->lib_h_ready = false;        // (the half copy for the packed-fp16 pre-tier is made by the first call that wants it)
    e->pre_auto = mm_engine::PreAuto{};
    // whole-image statistics for stage 0 of the filter's bound (the blocks holding active pixels, joined), into slot ppad / 64
    for (int v = 0; v < n_var; ++v)
        HIP_TRY(e, mm::launch_image_stats((float*)e->lib_lsum.p + (size_t)v * L * stat_slots(ppad), (size_t)n_var * L * stat_slots(ppad),
                                          stat_slots(ppad), ppad / 64, (e->var_active[v] + 63) / 64, (size_t)L, e->stream),
                "Error during library image-statistics kernel");
    // ... and the three channel sums per block once more, candidate-fastest, for the Euclid filter's stage 1
    if ((st = ensure(e, e->lib_stats_t, (size_t)n_var * 3 * (ppad / 64) * L * sizeof(float), "Failed to allocate transposed block sums"))) return st;
    for (int v = 0; v < n_var; ++v)
        HIP_TRY(e, mm::launch_transpose_stats((const float*)e->lib_lsum.p + (size_t)v * L * stat_slots(ppad), (size_t)n_var * L * stat_slots(ppad),
                                              stat_slots(ppad), ppad / 64, (float*)e->lib_stats_t.p + (size_t)v * 3 * (ppad / 64) * L, (size_t)L,
                                              e->stream), "Error during block-sum transpose kernel");
    if (P <= mm::kTinyPositions) {
        // position-major copy for tiny_cells_kernel: [variant][position < 64][channel][image]
        const size_t n = (size_t)n_var * mm::kTinyPositions * 3 * L;
        if ((st = ensure(e, e->lib_tiny, n * sizeof(float), "Failed to allocate the position-major library"))) return st;
        mm::transpose_tiny_kernel<<<(unsigned)std::min<size_t>((n + mm::kBlock - 1) / mm::kBlock, 65536), mm::kBlock, 0, e->stream>>>(
            (const float*)e->lib.p, (float*)e->lib_tiny.p, n_var, L, ppad, mm::kTinyPositions);
        HIP_TRY(e, hipGetLastError(), "Error during library transpose kernel");
    }
    HIP_TRY(e, sync_stream(e), "Failed to synchronise stream");   // host vectors go out of scope
    e->planes_dirty = false;
    return MM_OK;
}

mm::Variants variants_of(const mm_engine* e)
{
    const int P = e->lib_size * e->lib_size, ppad = round_up(P, 256);
    mm::Variants v{};
    v.cell_var = (const uint8_t*)e->cell_var.p;
    v.lib_stride = (size_t)e->n_lib * 3 * ppad;
    v.chroma_stride = (size_t)e->n_lib * ppad;
    for (int i = 0; i < 4; ++i) v.groups[i] = e->var_groups[i];
    v.lib_stats = (const float*)e->lib_lsum.p;
    v.cell_stats = (const float*)e->cell_lsum.p;
    v.lib_stats_stride = (size_t)e->n_var * e->n_lib * stat_slots(ppad);
    v.cell_stats_stride = e->cell_lsum.bytes / sizeof(float) / 8;
    v.n_blocks = stat_slots(ppad);
    v.cell_full = (const uint8_t*)e->cell_full.p;
    v.two_stage = e->two_level ? 1 : 0;     // (bit 1, stage 0, is set per batch once the cells' image statistics exist)
    v.audit = (unsigned int)e->audit;
    v.pre_abs = (e->pretier_band != 0 ? mm::kPreBandAbsObserved : mm::kPreBandAbs) * 0.01f * (float)e->pre_band_pct;
    v.pre_rel = (e->pretier_band != 0 ? mm::kPreBandRelObserved : mm::kPreBandRel) * 0.01f * (float)e->pre_band_pct;
    v.bound_scale = 0.01f * (float)e->bound_pct;
    return v;
}

// The hue table of colour_diff_fast.hpp, in fp64 from the reference formulas: entry i is the
// mean hue whose diamond angle is t_i = 4 i / N; {T, G} with G = -2 sin(2 d_theta),
// d_theta = rad(30) exp(-((h - rad(275)) / rad(25))^2) and h in [0, 2 pi) NOT wrapped around
// 275 degrees (ColourDifference.cpp:124-133,145).
std::vector<float> build_hue_lut()
{
    constexpr int N = mm::kHueLutSize;
    constexpr double pi = 3.14159265358979323846;
    auto rad = [&](double d) { return d * pi / 180.0; };
    std::vector<double> T(N + 1), G(N + 1);
    for (int i = 0; i <= N; ++i) {
        const double t = 4.0 * (i % N) / N;
        // invert t = 2 - copysign(1 + q, hs), q = hc / (|hc| + |hs|)
        const bool upper = t <= 2.0;
        const double q = upper ? 1.0 - t : t - 3.0;
        const double hc = q, hs = (upper ? 1.0 : -1.0) * (1.0 - std::fabs(q));
        double h = std::atan2(hs, hc);
        if (h < 0.0) h += 2.0 * pi;
        T[i] = 1.0 - 0.17 * std::cos(h - rad(30.0)) + 0.24 * std::cos(2.0 * h) +
               0.32 * std::cos(3.0 * h + rad(6.0)) - 0.20 * std::cos(4.0 * h - rad(63.0));
        const double z = (h - rad(275.0)) / rad(25.0);
        G[i] = -2.0 * std::sin(2.0 * rad(30.0) * std::exp(-z * z));      // R_T = G * sqrt(C^7 / (C^7 + 25^7)), :136,:145
    }
    std::vector<float> lut((size_t)mm::kHueLutEntries * 2);       // entries 0 .. N + 1: N repeats 0, N + 1 repeats 1
    for (int i = 0; i < mm::kHueLutEntries; ++i) {
        lut[(size_t)2 * i + 0] = (float)(0.0075 * T[i % N]);       // S_H = 1 + 0.015 C' T = fma(C1' + C2', 0.0075 T, 1): one instruction
        lut[(size_t)2 * i + 1] = (float)G[i % N];
    }
    return lut;
}

mm_status ensure_hue_lut(mm_engine* e)
{
    if (e->hue_lut.p) return MM_OK;
    const std::vector<float> lut = build_hue_lut();
    mm_status st = ensure(e, e->hue_lut, lut.size() * sizeof(float), "Failed to allocate hue table");
    if (st) return st;
    // (a non-null table means a FILLED table -- also to the engine that adopts this one's buffers: a failed upload leaves none)
    const hipError_t code = hipMemcpy(e->hue_lut.p, lut.data(), lut.size() * sizeof(float), hipMemcpyHostToDevice);
    if (code != hipSuccess) { release(e->hue_lut); return hip_fail(e, code, "Failed to copy hue table to device", __FILE__, __LINE__); }
    return MM_OK;
}

// Error bounds of the fp32 filter forms (colour_diff_fast.hpp) against the fp64 forms,
// per pixel pair:  |fp32 - fp64| <= px_abs + px_rel * value, plus the fp32 accumulation of
// one lane's share of a cell (relative, grows with the pixels per lane).
// Measured on 2e8 random and adversarial pairs (tools/filter_accuracy.py,
// tests/test_gpu_filter.py): CIEDE2000 max |err| = 2.5e-4 + 1.8e-5 * value (the relative part
// is the linear interpolation of the hue table); the bound trusted here is 4x the absolute
// and 3x the relative part.  Euclid: three roundings of a value <= 442.
mm::FilterBand filter_band(int diff_type, int ppad, int percent = 100)
{
    mm::FilterBand b;
    b.px_abs = diff_type == MM_CIEDE2000 ? 1e-3f : 5e-4f;
    b.rel = (diff_type == MM_CIEDE2000 ? 6e-5f : 0.0f) + 2e-5f + 1.2e-7f * (float)(ppad / 64);
    if (percent != 100) {            // diagnostics only (mm_set_option "band_pct"): a deliberately wrong band, to see the guards fire
        b.px_abs *= 0.01f * (float)percent;
        b.rel *= 0.01f * (float)percent;
    }
    return b;
}

// fp64-only scoring of `b` cells whose records are at recs_dev: gather, score every
// (cell, candidate), keep the K lowest.  Results to ci / cs ([b][K]).
mm_status dense_batch(mm_engine* e, const mm_cell* recs_dev, int b, int diff_type, int K, int KW,
                      int32_t* ci, double* cs, const int* row_map_dev, hipEvent_t after_gather, hipEvent_t after_score)
{
    const int L = e->n_lib, size = e->lib_size, P = size * size, ppad = round_up(P, 256);
    const int cand_per_block = 16, cell_group = 512;
    mm_status stv = ensure(e, e->cell_var, (size_t)std::max(b, 1), "Failed to allocate cell variants");
    if (stv) return stv;
    if ((stv = ensure(e, e->cell_full, (size_t)std::max(b, 1), "Failed to allocate cell flags"))) return stv;
    const mm::Variants var = variants_of(e);
    const int4 vom = make_int4(e->var_of_mask[0], e->var_of_mask[1], e->var_of_mask[2], e->var_of_mask[3]);
    mm::gather_cells_kernel<<<dim3((ppad + mm::kBlock - 1) / mm::kBlock, b), mm::kBlock, 0, e->stream>>>(
        recs_dev, (const float*)e->target.p, e->rows, e->cols, cell_resize_of(e), (const uint8_t*)e->masks.p,
        (float*)e->cells.p, (uint8_t*)e->cmask.p, size, P, ppad, (const int*)e->perm.p, vom,
        (uint8_t*)e->cell_var.p, (uint8_t*)e->cell_full.p, nullptr);
    HIP_TRY(e, hipGetLastError(), "Error launching cell gather kernel");
    if (after_gather) HIP_TRY(e, hipEventRecord(after_gather, e->stream), "event");
    const long long n_tiles = (L + cand_per_block - 1) / cand_per_block;
    const long long blocks = (long long)b * n_tiles;
    by_type(diff_type, [&](auto T) {
        mm::score_kernel<decltype(T)::value><<<dim3((unsigned)std::min(blocks, mm::kMaxBlocks256)), mm::kBlock, 0, e->stream>>>(
            (const float*)e->cells.p, (const uint8_t*)e->cmask.p, (const float*)e->lib.p,
            (double*)e->scores.p, b, L, ppad, cand_per_block, cell_group, var);
        return MM_OK;
    });
    HIP_TRY(e, hipGetLastError(), "Error during colour difference / add reduction kernel");
    if (after_score) HIP_TRY(e, hipEventRecord(after_score, e->stream), "event");
    if (K >= L)
        mm::full_list_kernel<<<b, mm::kBlock, 0, e->stream>>>((const double*)e->scores.p, L, K, ci, cs, KW, row_map_dev);
    else
        mm::select_kernel<<<b, mm::kBlock, 0, e->stream>>>((const double*)e->scores.p, L, K, kTieBand, ci, cs, KW, row_map_dev);
    HIP_TRY(e, hipGetLastError(), "Error during candidate selection kernel");
    return MM_OK;
}

// Cells the filter could not vouch for (flags_host[i] != 0, or all of them after a discontinuity-log
// overflow) go through the fp64-only path; their lists land in their rows of the caller's arrays.
// `first` = index of flags_host[0] among the call's cells.
mm_status redo_flagged_cells(mm_engine* e, const mm_cell* cells_host, int first, int b, const uint8_t* flags_host, bool all_dense,
                             int diff_type, int K, int KW, int32_t* cand_i_dev, double* cand_s_dev, const int* row_map_host)
{
    std::vector<mm_cell> dense_cells;
    std::vector<int> dense_rows;
    for (int i = 0; i < b; ++i) {
        if (flags_host[(size_t)i] & 4) e->timing.band_alarms += 1;
        if (all_dense || flags_host[(size_t)i]) {
            dense_cells.push_back(cells_host[first + i]);
            dense_rows.push_back(row_map_host ? row_map_host[first + i] : first + i);
        }
    }
    if (dense_cells.empty()) return MM_OK;
    const int nd = (int)dense_cells.size();
    mm_status st;
    // the selection kernel writes each cell's list straight into its row of the caller's arrays (row map on the device)
    if ((st = ensure(e, e->dense_recs, (size_t)nd * sizeof(mm_cell), "Failed to allocate cell records")) ||
        (st = ensure(e, e->dense_i, (size_t)nd * sizeof(int), "Failed to allocate row map")))
        return st;
    HIP_TRY(e, hipMemcpyAsync(e->dense_recs.p, dense_cells.data(), (size_t)nd * sizeof(mm_cell),
                              hipMemcpyHostToDevice, e->stream), "Failed to copy cell records to device");
    HIP_TRY(e, hipMemcpyAsync(e->dense_i.p, dense_rows.data(), (size_t)nd * sizeof(int),
                              hipMemcpyHostToDevice, e->stream), "Failed to copy row map to device");
    HIP_TRY(e, hipEventRecord(e->ev[10], e->stream), "event");
    if ((st = dense_batch(e, (const mm_cell*)e->dense_recs.p, nd, diff_type, K, KW, cand_i_dev, cand_s_dev,
                          (const int*)e->dense_i.p, nullptr, nullptr)))
        return st;
    HIP_TRY(e, hipEventRecord(e->ev[11], e->stream), "event");
    HIP_TRY(e, sync_stream(e), "Failed to synchronise stream");
    e->timing.rescore_ms += elapsed(e->ev[10], e->ev[11]);
    e->timing.dense_cells += nd;
    return MM_OK;
}

// Phase times and work counters of a score batch whose events have completed.
void account_score_batch(mm_engine* e, bool two_tier, bool prune, int diff_type, int b, unsigned int edge_n,
                         const unsigned long long work[3], bool voided = false)
{
    if (!two_tier) {
        if (e->phase_events) {
            e->timing.stage_ms += elapsed(e->ev[0], e->ev[1]);
            e->timing.score_ms += elapsed(e->ev[1], e->ev[2]);
            e->timing.select_ms += elapsed(e->ev[2], e->ev[3]);
        }
        e->timing.score_launches += 1;
        return;
    }
    e->timing.visited_groups += (int64_t)work[0];
    e->timing.total_groups += (int64_t)work[1];
    e->timing.bound_dropped += (int64_t)work[2];
    if (e->pre_auto.pending >= 0) {             // a pre-tier trial: this launch's time, and the verdict once both forms have one
        mm_engine::PreAuto& a = e->pre_auto;
        if (!voided) {                          // (a voided launch's time is not a measurement of its form: the trial is taken again)
            a.ms[a.pending] = elapsed(e->ev[12], e->ev[13]);
            a.have |= 1 << a.pending;
        }
        a.pending = -1;
        if (a.have == 3) a.choice = a.ms[1] < 0.98f * a.ms[0] ? 1 : 0;
    }
    if (prune && diff_type == MM_CIEDE2000 && b > 0) {
        e->drop_rate = (double)work[2] / ((double)b * (double)e->n_lib);
        e->drop_rate_lib = e->n_lib; e->drop_rate_size = e->lib_size;
    }
    if (e->phase_events) {
        e->timing.stage_ms += elapsed(e->ev[0], e->ev[1]);
        e->timing.seed_ms += elapsed(e->ev[1], e->ev[4]);
        e->timing.score_ms += elapsed(e->ev[4], e->ev[2]);
        e->timing.audit_ms += elapsed(e->ev[2], e->ev[14]);
        e->timing.select_ms += elapsed(e->ev[14], e->ev[3]) + elapsed(e->ev[8], e->ev[9]);
        e->timing.rescore_ms += elapsed(e->ev[3], e->ev[8]);
    }
    e->timing.score_launches += 1;
    e->timing.edge_pairs += edge_n;
    e->timing.voided_launches += voided ? 1 : 0;
}

// Scores `n` cells in batches and leaves the K lowest candidates per cell in rows of
// cand_i_dev / cand_s_dev that are KW >= K entries wide (the rest of a row is emptied).
// Cell i goes to row row_map_host[i] (null: row i).  cells_host = the same records on the
// host (needed to re-run flagged cells through the fp64-only path).
// defer: when the cells fit ONE batch, everything is enqueued and the function returns WITHOUT asking the
// host anything (e->pend.active): the caller reads the cell flags and counters back later, in its own copy
// (mm_generate's single-submission path), and calls account_score_batch / redo_flagged_cells itself.
mm_status score_cells_device(mm_engine* e, const mm_cell* cells_host, const mm_cell* recs_dev, int n,
                             int diff_type, int K, int KW, int32_t* cand_i_dev, double* cand_s_dev,
                             const int* row_map_host,
                             mm_progress_cb progress, void* user, volatile int* cancel, bool defer = false)
{
    const int L = e->n_lib, size = e->lib_size, P = size * size, ppad = round_up(P, 256);
    e->pend.active = false;
    mm_status stp = prepare_planes(e);
    if (stp) return stp;
    const int* row_map_dev = nullptr;
    if (row_map_host) {
        if ((stp = ensure(e, e->row_map, (size_t)n * sizeof(int), "Failed to allocate row map"))) return stp;
        HIP_TRY(e, hipMemcpyAsync(e->row_map.p, row_map_host, (size_t)n * sizeof(int), hipMemcpyHostToDevice, e->stream),
                "Failed to copy row map to device");
        row_map_dev = (const int*)e->row_map.p;
    }
    // detail cells of up to 64 pixels: one wave per cell from the image to the list (tiny_cells_kernel), fp64 throughout
    const bool tiny = e->tiny_path != 0 && P <= mm::kTinyPositions && e->lib_tiny.p != nullptr;     // (prepare_planes above built it for this P)
    // (values outside the box the filter band was validated on -- possible only through the f32 setters -- go fp64-only)
    const bool in_range = filter_band_covers(e, diff_type);
    if (!in_range) e->timing.filter_off_range = 1;
    const bool two_tier = !tiny && e->use_filter != 0 && K < L && in_range;
    const int Kc = two_tier ? std::min(L, round_up(K + 32, 4)) : 0;
    // Early exit needs seeds: K <= M <= 256 of them.
    // (extra seeds tighten the CIEDE2000 limit -- 10..30 measured flat within 1 % of the filter time -- but buy the RGB /
    // CIE76 filter nothing: config B's kernel runs 0.315 ms at 0, 2, 6, 10 or 14 extra seeds while the seed phase grows from
    // 0.042 to 0.129 ms, so the Euclidean forms take two)
    // (and a SMALL CIEDE2000 call -- the reference harness's line, config H: 1280 cells of 25 x 25 against 214 images -- is made of
    // its fixed costs: 2 extra seeds instead of 14 take the seed phase from 0.109 to 0.041 ms and the step from 0.576 to 0.520 ms,
    // while config C, 27 times the work, runs the same within 1 % from 6 to 14; tools/experiments/archive/r4_third.sh)
    const bool small_call = (double)std::min(n, 32768) * L * P < 1e9;
    const int extra_seeds = e->extra_seeds >= 0 ? e->extra_seeds : (diff_type == MM_CIEDE2000 ? (small_call ? 2 : 14) : 2);
    const int M = std::min(L, std::min(256, round_up(K + extra_seeds, 4)));
    // Seeds and the early exit pay three launches (one when fused) and ~M full scores per cell for the pixel pairs they
    // save: below ~1e8 pairs of an RGB launch (~2e7 of CIEDE2000's, eight times the work per pair) the unpruned filter
    // is over before the seeds would be
    const double call_pairs = (double)std::min(n, 32768) * L * P;     // (irrelevant to tiny cells: they neither seed nor prune)
    const double min_pairs = e->prune_min_pairs >= 0 ? 1e3 * (double)e->prune_min_pairs : (diff_type == MM_CIEDE2000 ? 2e7 : 1e8);
    const bool prune = two_tier && e->use_prune != 0 && K <= M && call_pairs >= min_pairs;

    // Batch so that every per-cell workspace stays within a budget: score row, the planar / NaN-marked / chroma /
    // interleaved copies of the cell and its mask, block statistics, contenders, seeds, flags.
    const size_t per_cell = tiny ? (size_t)L * sizeof(double) + 16
                                 : (size_t)L * sizeof(double) + (size_t)ppad * (12 + 1 + (two_tier ? 4 + 4 + 16 + 8 : 0)) +
                                   (size_t)8 * stat_slots(ppad) * sizeof(float) + (size_t)Kc * 16 + (prune ? (size_t)M * 8 + 4 : 0) + 20;
    // hipMemGetInfo is a driver round trip: asked again only when this call needs more cells per batch than the
    // last budget allowed (the workspaces below are kept between calls)
    // (a batch is a grid dimension of the general path's kernels: 32768; tiny cells have no such bound, and a job of theirs
    // is hundreds of thousands of cells -- 312 500 for the reference harness's cell size 8 on a 5000 x 4000 image)
    const int cell_cap = tiny ? (1 << 22) : 32768;
    const int want_cells = std::min(n, cell_cap);
    if (e->batch_per_cell != per_cell || e->batch_capacity < want_cells) {
        size_t free_b = 0, total_b = 0;
        HIP_TRY(e, hipMemGetInfo(&free_b, &total_b), "Failed to query device memory");
        const size_t held = e->scores.bytes + e->cells.bytes + e->cmask.bytes + e->cells_il.bytes + e->cells_f0.bytes + e->cell_chroma.bytes;
        const size_t budget = std::min<size_t>(free_b / 2 + held, (size_t)6 << 30);
        if (budget < per_cell) {
            set_err(e, "Not enough device memory: one cell needs %zu bytes of workspace, %zu free",
                    per_cell, free_b);
            return MM_OUT_OF_MEMORY;
        }
        e->batch_per_cell = per_cell;
        e->batch_capacity = (int)std::min<size_t>(budget / per_cell, (size_t)cell_cap);
    }
    int batch = std::min(n, e->batch_capacity);
    if (e->max_batch_cells > 0) batch = std::min(batch, e->max_batch_cells);
    unsigned int edge_cap = 0;
    // Workspaces of a batch; a refused allocation (memory taken since the capacity was computed: a larger target, a
    // second engine on the device) halves the batch instead of failing the call.
    auto allocate = [&](int bt) -> mm_status {
        mm_status st;
        if (tiny) return ensure(e, e->scores, (size_t)bt * L * sizeof(double), "Failed to allocate variants");
        if ((st = ensure(e, e->cell_var, (size_t)bt, "Failed to allocate cell variants")) ||
            (st = ensure(e, e->cell_full, (size_t)bt, "Failed to allocate cell flags")) ||
            (st = ensure(e, e->cell_order, (size_t)bt * sizeof(int32_t), "Failed to allocate the cell order")) ||
            (st = ensure(e, e->cell_lsum, (size_t)8 * bt * stat_slots(ppad) * sizeof(float), "Failed to allocate block statistics")) ||
            (st = ensure(e, e->cells, (size_t)bt * 3 * ppad * sizeof(float), "Failed to allocate cell images")) ||
            (st = ensure(e, e->cmask, (size_t)bt * ppad, "Failed to allocate cell masks")) ||
            (st = ensure(e, e->scores, (size_t)bt * L * sizeof(double), "Failed to allocate variants")))
            return st;
        if (two_tier) {
            const double pairs = (double)bt * L * P;
            edge_cap = (unsigned int)std::min<double>(64e6, std::max<double>(1e6, pairs / 2000.0));
            if ((st = ensure(e, e->cont_i, (size_t)bt * Kc * sizeof(int32_t), "Failed to allocate contenders")) ||
                (st = ensure(e, e->cont_f, (size_t)bt * Kc * sizeof(float), "Failed to allocate contenders")) ||
                (st = ensure(e, e->cont_s, (size_t)bt * Kc * sizeof(double), "Failed to allocate contenders")) ||
                (st = ensure(e, e->flags, (size_t)bt, "Failed to allocate cell flags")) ||
                (st = ensure(e, e->sentinel, (size_t)bt * sizeof(int32_t), "Failed to allocate band guard slots")) ||
                (st = ensure(e, e->edge_count, mm::kFilterCounterWords * sizeof(unsigned int), "Failed to allocate edge counter")) ||
                (st = ensure(e, e->edges, (size_t)edge_cap * sizeof(mm::EdgeEntry), "Failed to allocate edge list")) ||
                (st = ensure_hue_lut(e)) ||
                (st = ensure(e, e->cell_chroma, (size_t)bt * ppad * sizeof(float), "Failed to allocate cell chroma planes")) ||
                (st = ensure(e, e->cells_f0, (size_t)bt * ppad * sizeof(float), "Failed to allocate cell planes")) ||
                (st = ensure(e, e->cells_il, (size_t)bt * 4 * ppad * sizeof(float), "Failed to allocate interleaved cells")) ||
                (MM_WITH_PRETIER != 0 && (st = ensure(e, e->cells_h, (size_t)bt * 4 * ppad * sizeof(uint16_t), "Failed to allocate half-precision cells"))))
                return st;
        }
        if (prune) {
            if ((st = ensure(e, e->seeds, (size_t)bt * M * sizeof(int32_t), "Failed to allocate seeds")) ||
                (st = ensure(e, e->seed_score, (size_t)bt * M * sizeof(float), "Failed to allocate seeds")) ||
                (st = ensure(e, e->limits, (size_t)bt * sizeof(float), "Failed to allocate pruning limits")))
                return st;
        }
        return MM_OK;
    };
    mm_status st;
    while ((st = allocate(batch)) == MM_OUT_OF_MEMORY && batch > 1) {
        for (DevBuf* bf : { &e->scores, &e->cells, &e->cmask, &e->cells_il, &e->cells_f0, &e->cell_chroma, &e->cont_s, &e->edges })
            release(*bf);
        batch = (batch + 1) / 2;
        e->batch_capacity = batch;
    }
    if (st) return st;

    const mm::FilterBand band = filter_band(diff_type, ppad, e->band_pct);
    const int ftype = diff_type == MM_CIEDE2000 ? mm::CIEDE2000 : mm::RGB_EUCLIDEAN;
    mm::FilterInputs fin{};
    fin.cells = (const float*)e->cells.p; fin.cmask = (const uint8_t*)e->cmask.p;
    fin.cells_f0 = (const float*)e->cells_f0.p;
    fin.cell_chroma = (const float*)e->cell_chroma.p; fin.lib = (const float*)e->lib.p;
    fin.lib_chroma = (const float*)e->lib_chroma.p; fin.hue_lut = (const float2*)e->hue_lut.p;
    fin.var = variants_of(e);
    fin.tile_candidates = e->tile_candidates;
    fin.lds_pad_bytes = (size_t)e->lds_pad_kb * 1024;
    fin.use_lds_form = e->use_lds_filter < 0 ? (diff_type != MM_CIEDE2000) : e->use_lds_filter;
    fin.lib_il = e->use_interleave && diff_type == MM_CIEDE2000 ? (const float*)e->lib_il.p : nullptr;
    fin.cells_il = e->use_interleave && diff_type == MM_CIEDE2000 ? (const float*)e->cells_il.p : nullptr;
    fin.sort_groups = e->sort_groups;
    fin.xcd_aware = e->xcd_aware;
    fin.xcd_cells = e->xcd_cells;
    fin.item_chunk = e->item_chunk;
    fin.check_every = e->check_every;
    fin.cell_order = e->long_items_first && e->any_clipped ? (int32_t*)e->cell_order.p : nullptr;     // (no clipped cell: nothing to move)
    fin.lib_stats_t = e->euclid_stage1 != 0 && diff_type != MM_CIEDE2000 ? (const float*)e->lib_stats_t.p : nullptr;
    const int4 vom = make_int4(e->var_of_mask[0], e->var_of_mask[1], e->var_of_mask[2], e->var_of_mask[3]);
    // the fused ranking kernel keeps a cell's contenders in LDS (16 bytes each)
    // (config D, 4096 cells of CIEDE2000: the separate re-score balances the contenders better -- 2.21 + 0.15 ms against
    // 2.53 fused; up to ~2000 cells the two launches saved weigh more)
    // ... and only for SHORT lists: a workgroup re-scores its cell's contenders four at a time, so a deepened cell (K = 42 at
    // repeat range 4: >= 42 contenders of 4096 pixels each) keeps one workgroup busy for eleven rounds of fp64 CIEDE2000 while the
    // device idles -- photographic content, where neighbours collide and a fifth of the cells are deepened (bench.py --config P):
    // 5.96 ms of an 11.2 ms step in this kernel; the separate re-score gives every (cell, contender) a wave of its own
    const bool fused_rank = e->fused != 0 && (size_t)Kc * 16 <= 48 * 1024 && (diff_type != MM_CIEDE2000 || n <= 2048) &&
                            (K <= 8 || diff_type != MM_CIEDE2000);

    std::vector<uint8_t> flags_host;
    for (int done = 0; done < n; done += batch) {
        if (cancel && *cancel) return MM_CANCELLED;
        const int b = std::min(batch, n - done);
        int32_t* ci = row_map_dev ? cand_i_dev : cand_i_dev + (size_t)done * KW;
        double* cs = row_map_dev ? cand_s_dev : cand_s_dev + (size_t)done * KW;
        const int* rmap = row_map_dev ? row_map_dev + done : nullptr;
        const bool deferred = defer && n <= batch;
        // single submission: the filter's counter block is the one inside the result header (zeroed by result_init_kernel)
        mm::ResultHeader* const hdr = deferred ? e->fast_hdr : nullptr;
        unsigned int* const counters_dev = hdr ? hdr->filter_counters : (unsigned int*)e->edge_count.p;
        fin.counters_cleared = hdr ? 1 : 0;
        // one launch for seeds + scores + limit where the batch is small: measured (round 3) config B 0.147 -> 0.129 ms, but
        // config C (1694 cells of CIEDE2000, whose seed scores carry the fp64 form inline) 0.32 -> 0.47 ms: the separate
        // score kernel spreads the (cell, seed) pairs over the device, the fused one keeps a cell's seeds on one CU
        fin.fused_seeds = e->fused_seed_cells > 0 && b <= e->fused_seed_cells && (diff_type != MM_CIEDE2000 || (long long)b * M <= 4096);
        if (e->phase_events) HIP_TRY(e, hipEventRecord(e->ev[0], e->stream), "event");
        if (tiny) {
            if (e->phase_events) HIP_TRY(e, hipEventRecord(e->ev[1], e->stream), "event");
            by_type(diff_type, [&](auto T) {
                mm::tiny_cells_kernel<decltype(T)::value><<<(unsigned)((b + 3) / 4), mm::kBlock, 0, e->stream>>>(
                    recs_dev + done, b, (const float*)e->target.p, e->rows, e->cols, cell_resize_of(e), (const uint8_t*)e->masks.p,
                    size, ppad, (const int*)e->perm.p, vom, (const float*)e->lib_tiny.p, L, (double*)e->scores.p,
                    K, kTieBand, ci, cs, KW, rmap);
                return MM_OK;
            });
            HIP_TRY(e, hipGetLastError(), "Error during tiny-cell kernel");
            if (e->phase_events) {
                HIP_TRY(e, hipEventRecord(e->ev[2], e->stream), "event");
                HIP_TRY(e, hipEventRecord(e->ev[3], e->stream), "event");
            }
            if (deferred) {
                e->pend = mm_engine::Pending{true, false, false, false, b, diff_type, 0u};
                return MM_OK;
            }
            HIP_TRY(e, sync_stream(e), "Failed to synchronise stream");
            account_score_batch(e, false, false, diff_type, b, 0u, nullptr);
            if (progress) progress(done + b, user);
            continue;
        }
        if (!two_tier) {
            if ((st = dense_batch(e, recs_dev + done, b, diff_type, K, KW, ci, cs, rmap, e->phase_events ? e->ev[1] : nullptr, e->phase_events ? e->ev[2] : nullptr))) return st;
            if (e->phase_events) HIP_TRY(e, hipEventRecord(e->ev[3], e->stream), "event");
            if (deferred) {
                e->pend = mm_engine::Pending{true, false, false, false, b, diff_type, 0u};
                return MM_OK;
            }
            HIP_TRY(e, sync_stream(e), "Failed to synchronise stream");
            account_score_batch(e, false, false, diff_type, b, 0u, nullptr);
            if (progress) progress(done + b, user);
            continue;
        }

        if (e->fused != 0) {
            // one launch: planes, mask, NaN-marked channel 0, chroma plane, interleaved copy, block statistics
            const dim3 grid((unsigned)(ppad / mm::kBlock), (unsigned)b);
            const size_t stats_stride = e->cell_lsum.bytes / sizeof(float) / 8;
            if (diff_type == MM_CIEDE2000)
                mm::gather_prepare_kernel<true><<<grid, mm::kBlock, 0, e->stream>>>(
                    recs_dev + done, (const float*)e->target.p, e->rows, e->cols, cell_resize_of(e), (const uint8_t*)e->masks.p,
                    (float*)e->cells.p, (uint8_t*)e->cmask.p, size, P, ppad, (const int*)e->perm.p, vom,
                    (uint8_t*)e->cell_var.p, (uint8_t*)e->cell_full.p, (float*)e->cells_f0.p, (float*)e->cell_chroma.p,
                    e->use_interleave ? (float*)e->cells_il.p : nullptr, (float*)e->cell_lsum.p, stats_stride, stat_slots(ppad));
            else
                mm::gather_prepare_kernel<false><<<grid, mm::kBlock, 0, e->stream>>>(
                    recs_dev + done, (const float*)e->target.p, e->rows, e->cols, cell_resize_of(e), (const uint8_t*)e->masks.p,
                    (float*)e->cells.p, (uint8_t*)e->cmask.p, size, P, ppad, (const int*)e->perm.p, vom,
                    (uint8_t*)e->cell_var.p, (uint8_t*)e->cell_full.p, (float*)e->cells_f0.p, nullptr, nullptr,
                    (float*)e->cell_lsum.p, stats_stride, stat_slots(ppad));
            HIP_TRY(e, hipGetLastError(), "Error launching cell gather kernel");
        } else {
            mm::gather_cells_kernel<<<dim3((ppad + mm::kBlock - 1) / mm::kBlock, b), mm::kBlock, 0, e->stream>>>(
                recs_dev + done, (const float*)e->target.p, e->rows, e->cols, cell_resize_of(e), (const uint8_t*)e->masks.p,
                (float*)e->cells.p, (uint8_t*)e->cmask.p, size, P, ppad, (const int*)e->perm.p, vom,
                (uint8_t*)e->cell_var.p, (uint8_t*)e->cell_full.p, (float*)e->cells_f0.p);
            HIP_TRY(e, hipGetLastError(), "Error launching cell gather kernel");
            if (diff_type == MM_CIEDE2000) {
                HIP_TRY(e, mm::launch_chroma_planes((const float*)e->cells.p, (float*)e->cell_chroma.p, (size_t)b, ppad, e->stream),
                        "Error during cell chroma kernel");
                if (e->use_interleave)
                    HIP_TRY(e, mm::launch_interleave((const float*)e->cells.p, (const float*)e->cell_chroma.p, (const float*)e->cells_f0.p,
                                                     (float*)e->cells_il.p, (size_t)b, ppad, e->stream), "Error during cell interleave kernel");
            }
            HIP_TRY(e, mm::launch_group_sums((const float*)e->cells.p, (const uint8_t*)e->cmask.p, (size_t)ppad,
                                             (float*)e->cell_lsum.p, e->cell_lsum.bytes / sizeof(float) / 8,
                                             (size_t)b, ppad, stat_slots(ppad), e->stream),
                    "Error during cell group-sum kernel");
        }
        // stage 0 of the bound wants the cells' statistics joined over the whole image too (one thread per cell)
        const bool stage0 = e->stage0 != 0 && prune && diff_type == MM_CIEDE2000;
        if (stage0)
            HIP_TRY(e, mm::launch_image_stats((float*)e->cell_lsum.p, e->cell_lsum.bytes / sizeof(float) / 8, stat_slots(ppad), ppad / 64,
                                              (e->var_active[0] + 63) / 64, (size_t)b, e->stream, (const uint8_t*)e->cell_var.p,
                                              make_int4((e->var_active[0] + 63) / 64, (e->var_active[1] + 63) / 64,
                                                        (e->var_active[2] + 63) / 64, (e->var_active[3] + 63) / 64)),
                    "Error during cell image-statistics kernel");
        fin.var.two_stage = (e->two_level ? 1 : 0) | (stage0 ? 2 : 0);
        // the packed-fp16 pre-tier reads half copies of the cells and the library (CIEDE2000 with the early exit, interleaved form)
        const bool pre_possible = MM_WITH_PRETIER != 0 && prune && diff_type == MM_CIEDE2000 && e->use_interleave != 0 && fin.use_lds_form == 0;
        bool pretier = pre_possible && e->pretier > 0;
        bool pre_trial = false;
        // (eligible by size -- small jobs lose to it, config C +5 % -- and not a deepening sub-batch: decided by measurement.  On the
        // worst-case band NO benchmarked configuration gains -- configs D, E, I lose 5 - 9 % to it, the 128-px cells of config G gained 2 %
        // until their item size was retuned and lose 5 % since (profiles/r05_pretier_band_ab.txt, HISTORY.md) -- so a trial would only
        // cost its two launches: on that band the pre-tier runs where it is asked for ("pretier" 1); the measured rule is the observed
        // band's, with round 4's eligibility)
        if (pre_possible && e->pretier < 0 && row_map_host == nullptr && e->pretier_band != 0 &&
            (ppad / 256 > 32 || call_pairs >= 1e10)) {
            // The state is keyed on the FULL batch of the job (library, cell size, cells per full batch) and only full batches touch it:
            // a job cut into several batches ends on a shorter one, which must neither reset what the full ones measured nor be timed
            // against them -- it runs the form chosen so far (none yet: without the pre-tier).  The two trials are the first two full
            // batches, of one call or of two.
            mm_engine::PreAuto& a = e->pre_auto;
            const bool full = b == batch;
            if (full && (a.lib != L || a.size != size || a.cells != batch)) { a = mm_engine::PreAuto{}; a.lib = L; a.size = size; a.cells = batch; }
            const bool mine = a.lib == L && a.size == size && a.cells == batch;
            // (the first full batch over a library runs cold -- first touch of the interleaved copies, the kernel's first launch --:
            // timed, it made the form tried second look 5 % better than it is and 80-px cells kept a pre-tier that lost; so it is a
            // warm-up without the pre-tier, then one launch without and one with, both timed)
            if (mine && a.choice >= 0) pretier = a.choice != 0;
            else if (mine && full && !a.warm) { pretier = false; a.warm = true; }
            // (the trial is PENDING only once its launch has gone out -- below, after launch_filter -- so that an error or a cancel between
            // here and there cannot leave a pending trial whose two events were never recorded: ADVICE round 5)
            else if (mine && full && a.pending < 0) { pretier = (a.have & 1) != 0; pre_trial = true; }    // off first, then on
            else pretier = false;
        }
        if (row_map_host == nullptr) e->timing.pretier = pretier ? 1 : 0;
        if (pretier && !e->lib_h_ready) {       // the library's half copy, once per library / mask change
            if ((st = ensure(e, e->lib_h, (size_t)e->n_var * L * 4 * ppad * sizeof(uint16_t), "Failed to allocate the half-precision library"))) return st;
            HIP_TRY(e, mm::launch_interleave_h((const float*)e->lib_il.p, e->lib_h.p, (size_t)e->n_var * L, ppad, e->stream),
                    "Error during half-precision library kernel");
            e->lib_h_ready = true;
        }
        fin.lib_h = pretier ? e->lib_h.p : nullptr;
        fin.cells_h = pretier ? e->cells_h.p : nullptr;
        // the observed band is a statistical guarantee: where it is in force the audit of the dropped candidates runs at eight times its rate
        fin.var.audit = pretier && e->pretier_band != 0 && e->audit > 8 ? (unsigned int)(e->audit / 8) : (unsigned int)e->audit;
        if (pretier)
            HIP_TRY(e, mm::launch_interleave_h((const float*)e->cells_il.p, e->cells_h.p, (size_t)b, ppad, e->stream),
                    "Error during half-precision cell kernel");
        if (e->phase_events) HIP_TRY(e, hipEventRecord(e->ev[1], e->stream), "event");

        // tier 1: fp32 filter score of every (cell, candidate), with early exit against a
        // per-cell limit obtained from a few proxy-chosen seeds
        if (prune)
            HIP_TRY(e, mm::launch_prune_limits(ftype, fin, (const float*)e->lib_means.p, b, L, ppad,
                                               M, K, band.px_abs, band.rel, (int32_t*)e->seeds.p,
                                               (float*)e->seed_score.p, (float*)e->limits.p, (float*)e->scores.p, e->stream),
                    "Error during seed / pruning limit kernels");
        if (e->phase_events) HIP_TRY(e, hipEventRecord(e->ev[4], e->stream), "event");
        fin.seeds = prune ? (const int32_t*)e->seeds.p : nullptr;
        fin.n_seeds = prune ? M : 0;
        // Item size by what the bound did last time on this library: where it drops nearly everything (photograph-like
        // content: 99 % at stage 1) an item is over in microseconds and its fixed latencies set the pace -- 64-candidate
        // items then run the filter in 0.6 of the time of 32-candidate ones; where half of the candidates survive (the
        // synthetic benchmark) both run the same and the smaller item moves less data.  Results do not depend on it;
        // the TIME of a call therefore depends on the engine's previous CIEDE2000 call over the same library
        // (mm_set_option("item_auto", 0) turns the feedback off; mm_timing.filter_tile reports what was used).
        fin.tile_candidates = e->tile_candidates;
        // (cells above 32 groups keep their measured 16-candidate items: the 64-candidate rule was measured on 64-px cells, and such
        // cells' launches are bound by the fabric, where FEWER cells and candidates in flight is what helps -- ADVICE round 5)
        if (e->tile_candidates == 0 && e->item_auto != 0 && prune && diff_type == MM_CIEDE2000 && e->drop_rate >= 0.9 &&
            e->drop_rate_lib == L && e->drop_rate_size == size && ppad / 256 <= 32)
            fin.tile_candidates = 64;
        int tile_used = 0;
        // (a trial's two events are recorded by launch_filter itself, directly around the kernel: after the one-time occupancy query
        // and attribute calls of a kernel's first launch, which are host time the device would otherwise be seen waiting for)
        fin.clock_begin = pre_trial ? e->ev[12] : nullptr;
        fin.clock_end = pre_trial ? e->ev[13] : nullptr;
        HIP_TRY(e, mm::launch_filter(ftype, fin, (float*)e->scores.p, b, L, ppad,
                                     (mm::EdgeEntry*)e->edges.p, counters_dev, edge_cap,
                                     prune ? (const float*)e->limits.p : nullptr, e->stream, &tile_used),
                "Error during filter colour difference kernel");
        if (pre_trial) e->pre_auto.pending = pretier ? 1 : 0;
        e->timing.filter_tile = tile_used;
        if (e->phase_events) HIP_TRY(e, hipEventRecord(e->ev[2], e->stream), "event");
        // the audit of what the filter dropped: reads the score rows, may void the launch through the counter block's void word
        HIP_TRY(e, mm::launch_filter_audit(ftype, fin, (const float*)e->scores.p, b, L, ppad, prune ? (const float*)e->limits.p : nullptr,
                                           counters_dev, e->stream),
                "Error during filter audit kernel");
        if (e->phase_events) HIP_TRY(e, hipEventRecord(e->ev[14], e->stream), "event");

        if (fused_rank) {
            // contenders, their fp64 reference-form scores and the K lowest of them in one launch
            if (e->phase_events) HIP_TRY(e, hipEventRecord(e->ev[3], e->stream), "event");
            by_type(diff_type, [&](auto T) {
                mm::rank_kernel<decltype(T)::value><<<b, mm::kBlock, (size_t)Kc * 16, e->stream>>>(
                    (const float*)e->scores.p, (const float*)e->cells.p, (const uint8_t*)e->cmask.p, (const float*)e->lib.p,
                    L, ppad, K, Kc, band, kTieBand, ci, cs, (uint8_t*)e->flags.p, KW, rmap, e->band_guard_every, fin.var,
                    hdr, K == KW ? 1 : 0);
                return MM_OK;
            });
            HIP_TRY(e, hipGetLastError(), "Error during contender ranking kernel");
            if (e->phase_events) HIP_TRY(e, hipEventRecord(e->ev[8], e->stream), "event");
            if (e->phase_events) HIP_TRY(e, hipEventRecord(e->ev[9], e->stream), "event");
        } else {
            mm::contenders_kernel<<<b, mm::kBlock, 0, e->stream>>>(
                (const float*)e->scores.p, (const uint8_t*)e->cmask.p, L, ppad, K, Kc, band,
                (int32_t*)e->cont_i.p, (float*)e->cont_f.p, (uint8_t*)e->flags.p, (int32_t*)e->sentinel.p, e->band_guard_every);
            HIP_TRY(e, hipGetLastError(), "Error during contender selection kernel");
            if (e->phase_events) HIP_TRY(e, hipEventRecord(e->ev[3], e->stream), "event");

            // tier 2: fp64 reference-form score of the contenders only
            const long long waves = (long long)b * Kc;
            by_type(diff_type, [&](auto T) {
                mm::rescore_kernel<decltype(T)::value><<<(unsigned)((waves + 3) / 4), mm::kBlock, 0, e->stream>>>(
                    (const float*)e->cells.p, (const uint8_t*)e->cmask.p, (const float*)e->lib.p,
                    (const int32_t*)e->cont_i.p, (double*)e->cont_s.p, b, Kc, ppad, fin.var);
                return MM_OK;
            });
            HIP_TRY(e, hipGetLastError(), "Error during contender re-score kernel");
            if (e->phase_events) HIP_TRY(e, hipEventRecord(e->ev[8], e->stream), "event");

            mm::rerank_kernel<<<b, mm::kBlock, 0, e->stream>>>(
                (const int32_t*)e->cont_i.p, (const float*)e->cont_f.p, (const double*)e->cont_s.p,
                (const uint8_t*)e->cmask.p, Kc, K, ppad, band, kTieBand, ci, cs, (uint8_t*)e->flags.p, KW, rmap,
                (const int32_t*)e->sentinel.p, L);
            HIP_TRY(e, hipGetLastError(), "Error during re-rank kernel");
            if (e->phase_events) HIP_TRY(e, hipEventRecord(e->ev[9], e->stream), "event");
        }
        if (deferred) {
            e->pend = mm_engine::Pending{true, true, prune, fused_rank && hdr != nullptr, b, diff_type, edge_cap};
            return MM_OK;
        }

        flags_host.resize((size_t)b);
        unsigned int counters[8] = {0};          // [0] logged discontinuity pairs, [1] the void word, [2..7] three u64 work counters
        HIP_TRY(e, hipMemcpyAsync(flags_host.data(), e->flags.p, (size_t)b, hipMemcpyDeviceToHost, e->stream), "flags");
        HIP_TRY(e, hipMemcpyAsync(counters, e->edge_count.p, sizeof counters, hipMemcpyDeviceToHost, e->stream), "edge count");
        HIP_TRY(e, sync_stream(e), "Failed to synchronise stream");
        const unsigned int edge_n = counters[0];
        const bool voided = counters[mm::kFilterVoidWord] != 0u;      // the pre-tier's canary or the audit: no score of the launch is trusted
        unsigned long long work[3];
        memcpy(work, counters + 2, sizeof work);
        account_score_batch(e, true, prune, diff_type, b, edge_n, work, voided);
        // cells the filter could not vouch for go through the fp64-only path
        if ((st = redo_flagged_cells(e, cells_host, done, b, flags_host.data(), voided || edge_n > edge_cap /* log overflowed */,
                                     diff_type, K, KW, cand_i_dev, cand_s_dev, row_map_host)))
            return st;
        if (progress) progress(done + b, user);
    }
    return MM_OK;
}

// The arguments every resolve kernel takes, from the engine's resident state.
mm::ResolveArgs resolve_args_of(const mm_engine* e, const mm_cell* recs_dev, int n, int grid_w, int grid_h,
                                int range, int addition, int K, const int32_t* cand_i_dev, const double* cand_s_dev,
                                long long* grid_dev, double* best_dev, const uint8_t* deep_dev, uint8_t* insufficient_dev, int k_fill)
{
    const int size = e->lib_size, P = size * size, ppad = round_up(P, 256);
    mm::ResolveArgs a{};
    a.recs = recs_dev; a.n_cells = n; a.grid_w = grid_w; a.grid_h = grid_h; a.L = e->n_lib;
    a.repeat_range = range; a.repeat_addition = addition; a.K = K;
    a.cand_index = cand_i_dev; a.cand_score = cand_s_dev;
    a.grid = grid_dev; a.best_score = best_dev;
    a.target = (const float*)e->target.p; a.rows = e->rows; a.cols = e->cols; a.cell_scale = cell_resize_of(e);
    a.masks = (const uint8_t*)e->masks.p; a.lib = (const float*)e->lib.p;
    a.lib_stride = (size_t)e->n_lib * 3 * ppad; a.inv_perm = (const int*)e->inv_perm.p;
    for (int i = 0; i < 4; ++i) a.var_of_mask[i] = e->var_of_mask[i];
    a.size = size; a.P = P; a.ppad = ppad;
    a.tie_band = kTieBand;
    a.deep = deep_dev; a.insufficient = insufficient_dev; a.k_fill = k_fill > 0 ? k_fill : K;
    return a;
}

// The wavefront forms of the raster-order dependency (kernels.hpp resolve_front_kernel / resolve_rows_kernel): ONE workgroup, the
// whole padded grid of picks in LDS.  Bytes of dynamic LDS they need, or 0 when the grid does not fit.
size_t wavefront_lds_bytes(int grid_w, int grid_h, bool front)
{
    // (the front form stages a row's totals and flags per wave: + kFrontWaves x grid_w x 9 bytes, rounded up)
    const size_t b = (((size_t)grid_w * grid_h + grid_h + 2) & ~(size_t)1) * sizeof(int) +
                     (size_t)(front ? mm::kFrontWaves : mm::kRowWaves) * mm::kRowExactChunk * sizeof(double) +
                     (front ? (size_t)mm::kFrontWaves * ((size_t)grid_w * 9 + 8) : 0);
    return b <= 158 * 1024 ? b : 0;
}

// Enqueues row index + wavefront kernel; `stuck` (device, zeroed by the caller or here) is raised if a wait timed out.
mm_status launch_wavefront(mm_engine* e, const mm::ResolveArgs& a, const mm_cell* recs_dev, int n, int grid_h, int diff_type,
                           bool front, size_t lds_bytes, int* stuck_dev)
{
    mm_status st;
    if ((st = ensure(e, e->row_index, ((size_t)2 * grid_h + 1) * sizeof(int), "Failed to allocate row index"))) return st;
    int* row_first = (int*)e->row_index.p;
    int* row_count = row_first + grid_h;
    int* stuck = stuck_dev ? stuck_dev : row_count + grid_h;
    HIP_TRY(e, hipMemsetAsync(row_first, 0, ((size_t)2 * grid_h + 1) * sizeof(int), e->stream), "Failed to clear row index");
    mm::row_index_kernel<<<std::min((n + mm::kBlock - 1) / mm::kBlock, 256), mm::kBlock, 0, e->stream>>>(recs_dev, n, row_first, row_count);
    HIP_TRY(e, hipGetLastError(), "Error during row index kernel");
    st = by_type(diff_type, [&](auto T) {
        auto kern = front ? mm::resolve_front_kernel<decltype(T)::value> : mm::resolve_rows_kernel<decltype(T)::value>;
        hipError_t code = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
        if (code != hipSuccess) return hip_fail(e, code, "Failed to raise the LDS limit", __FILE__, __LINE__);
        kern<<<1, (front ? mm::kFrontWaves : mm::kRowWaves) * mm::kWave, lds_bytes, e->stream>>>(a, row_first, row_count, stuck);
        return MM_OK;
    });
    if (st) return st;
    HIP_TRY(e, hipGetLastError(), "Error during calculate repeats / find lowest kernel");
    return MM_OK;
}

// deep_dev (may be null = every list is full): 1 where the list holds the full K; the
// others are speculative lists of k_fill entries whose pick is verified on the device;
// insufficient_dev receives the verdict per cell.
mm_status resolve_device(mm_engine* e, const mm_cell* recs_dev, int n, int grid_w, int grid_h,
                         int diff_type, int range, int addition, int K,
                         const int32_t* cand_i_dev, const double* cand_s_dev,
                         int64_t* best_fit_out, double* best_score_out,
                         const uint8_t* deep_dev = nullptr, uint8_t* insufficient_dev = nullptr, int k_fill = 0)
{
    const size_t cells_total = (size_t)grid_w * grid_h;
    mm_status st;
    if ((st = prepare_planes(e))) return st;
    if ((st = ensure(e, e->grid, cells_total * sizeof(long long), "Failed to allocate best fits")) ||
        (st = ensure(e, e->best, cells_total * sizeof(double), "Failed to allocate best scores")) ||
        (st = ensure(e, e->counter, 2 * sizeof(unsigned long long), "Failed to allocate counter")))
        return st;

    // Set best fits to max value (CUDAPhotomosaicGenerator.cpp:140-149), scores to NaN: one fill kernel, nothing
    // crosses PCIe and the host does not wait.
    const unsigned fill_blocks = (unsigned)std::min<size_t>((cells_total + mm::kBlock - 1) / mm::kBlock, 1024);
    auto fill_grid = [&]() -> hipError_t {
        mm::grid_fill_kernel<<<fill_blocks, mm::kBlock, 0, e->stream>>>((long long*)e->grid.p, (double*)e->best.p, cells_total, (long long)e->n_lib);
        return hipGetLastError();
    };
    HIP_TRY(e, fill_grid(), "Failed to copy max best fit from host to device");
    HIP_TRY(e, hipMemsetAsync(e->counter.p, 0, 2 * sizeof(unsigned long long), e->stream), "Failed to clear counter");

    mm::ResolveArgs a = resolve_args_of(e, recs_dev, n, grid_w, grid_h, range, addition, K, cand_i_dev, cand_s_dev,
                                        (long long*)e->grid.p, (double*)e->best.p, deep_dev, insufficient_dev, k_fill);
    a.exact_count = (unsigned long long*)e->counter.p;
    a.dense_count = a.exact_count + 1;

    const int sequential = (range > 0 && addition != 0) ? 1 : 0;
    const size_t ring_bytes = sequential ? (size_t)(range + 1) * grid_w * sizeof(int) : 0;
    const bool wave_form = e->use_wave_resolve != 0 && ring_bytes <= 120 * 1024;
    HIP_TRY(e, hipEventRecord(e->ev[4], e->stream), "event");
    if (n > 0 && wave_form) {
        // marked cells (list cut through a near-tie cluster): full rows, computed in parallel
        if ((st = ensure(e, e->row_slot, (size_t)n * sizeof(int) + sizeof(int), "Failed to allocate row slots"))) return st;
        int* slot = (int*)e->row_slot.p;
        int* mark_count = slot + n;
        HIP_TRY(e, hipMemsetAsync(mark_count, 0, sizeof(int), e->stream), "Failed to clear counter");
        mm::mark_scan_kernel<<<std::min((n + mm::kBlock - 1) / mm::kBlock, 256), mm::kBlock, 0, e->stream>>>(
            cand_i_dev, n, K, deep_dev, slot, mark_count);
        HIP_TRY(e, hipGetLastError(), "Error during mark scan kernel");
        int marked = 0;
        HIP_TRY(e, hipMemcpyAsync(&marked, mark_count, sizeof(int), hipMemcpyDeviceToHost, e->stream), "mark count");
        HIP_TRY(e, sync_stream(e), "Failed to synchronise stream");
        a.row_slot = marked > 0 ? slot : nullptr;      // no marks: skip the per-cell lookup
        if (marked > 0) {
            if ((st = ensure(e, e->dense_rows_buf, (size_t)marked * e->n_lib * sizeof(double), "Failed to allocate score rows")))
                return st;
            a.dense_rows = (const double*)e->dense_rows_buf.p;
            by_type(diff_type, [&](auto T) {
                mm::dense_rows_kernel<decltype(T)::value><<<n, mm::kBlock, 0, e->stream>>>(a, (double*)e->dense_rows_buf.p);
                return MM_OK;
            });
            HIP_TRY(e, hipGetLastError(), "Error during dense rows kernel");
        }
        // Repeat penalty as parallel sweeps to the raster-order fix point (resolve_wave_kernel,
        // mode 2): launched a few at a time, each with its own change counter; the first sweep
        // that changed nothing proves the state final.  Past max_sweeps (long collision chains,
        // e.g. a flat target) the grid is reset and the sequential forms below take over.
        bool swept = false;
        if (sequential && e->use_sweep_resolve != 0) {
            // (with the front form behind them the sweeps get two sets: content they do not settle in twelve passes is the front form's)
            const bool front_ready = e->use_front_resolve != 0 && wavefront_lds_bytes(grid_w, grid_h, true) != 0;
            const bool skip_sweeps = front_ready && e->front_learned && e->sweeps_lib == e->n_lib && e->sweeps_size == e->lib_size;
            const int max_sweeps = skip_sweeps ? 0 : (front_ready ? std::min(e->max_sweeps, 12) : e->max_sweeps), batch = 6;
            // per sweep: [max_sweeps] exact re-sum counters (u64), then [max_sweeps] change counters (u32)
            const size_t counts_bytes = (size_t)max_sweeps * (sizeof(unsigned long long) + sizeof(unsigned int));
            if ((st = ensure(e, e->sweep_counts, counts_bytes, "Failed to allocate sweep counters"))) return st;
            unsigned long long* exacts = (unsigned long long*)e->sweep_counts.p;
            unsigned int* counts = (unsigned int*)(exacts + max_sweeps);
            HIP_TRY(e, hipMemsetAsync(e->sweep_counts.p, 0, counts_bytes, e->stream), "Failed to clear sweep counters");
            const int blocks = std::max(1, std::min(n, 8192));
            std::vector<unsigned int> host_counts(batch);
            unsigned long long* const exact_total_count = a.exact_count;
            int launched = 0;
            while (!swept && launched < max_sweeps) {
                const int now = std::min(batch, max_sweeps - launched);
                for (int i = 0; i < now; ++i) {
                    a.sweep_changes = counts + launched + i;
                    a.exact_count = exacts + launched + i;
                    by_type(diff_type, [&](auto T) {
                        mm::resolve_wave_kernel<decltype(T)::value><<<blocks, mm::kWave, 0, e->stream>>>(a, 2);
                        return MM_OK;
                    });
                }
                HIP_TRY(e, hipGetLastError(), "Error during calculate repeats / find lowest kernel");
                HIP_TRY(e, hipMemcpyAsync(host_counts.data(), counts + launched, (size_t)now * sizeof(unsigned int),
                                          hipMemcpyDeviceToHost, e->stream), "sweep counters");
                HIP_TRY(e, sync_stream(e), "Failed to synchronise stream");
                for (int i = 0; i < now && !swept; ++i)
                    if (host_counts[i] == 0) {
                        swept = true;
                        e->timing.resolve_sweeps += launched + i + 1;
                        // the re-sums of the final state = those of the sweep that changed nothing
                        HIP_TRY(e, hipMemcpyAsync(exact_total_count, exacts + launched + i, sizeof(unsigned long long),
                                                  hipMemcpyDeviceToDevice, e->stream), "exact counter");
                    }
                launched += now;
            }
            a.exact_count = exact_total_count;
            if (!swept) {                       // not converged: start the chain from a clean grid
                e->timing.resolve_sweeps += launched;
                HIP_TRY(e, fill_grid(), "Failed to copy max best fit from host to device");
            }
        }
        // whole grid + progress counters + per-wave exact buffers in LDS?
        const bool front = e->use_front_resolve != 0;
        const size_t rows_lds = wavefront_lds_bytes(grid_w, grid_h, front);
        if (swept) {
            // final state reached by the sweeps
        } else if (sequential && e->use_row_pipeline != 0 && rows_lds != 0) {
            if (e->use_sweep_resolve != 0 && front) { e->front_learned = true; e->sweeps_learned = 0; e->sweeps_lib = e->n_lib; e->sweeps_size = e->lib_size; }
            if ((st = launch_wavefront(e, a, recs_dev, n, grid_h, diff_type, front, rows_lds, nullptr))) return st;
            int stuck_host = 0;
            HIP_TRY(e, hipMemcpyAsync(&stuck_host, (int*)e->row_index.p + 2 * grid_h, sizeof(int), hipMemcpyDeviceToHost, e->stream), "stuck flag");
            HIP_TRY(e, sync_stream(e), "Failed to synchronise stream");
            if (stuck_host) { set_err(e, "wavefront resolve made no progress (internal error)"); return MM_DEVICE_ERROR; }
        } else {
            const int blocks = sequential ? 1 : std::max(1, std::min(n, 8192));
            by_type(diff_type, [&](auto T) {
                mm::resolve_wave_kernel<decltype(T)::value><<<blocks, mm::kWave, ring_bytes, e->stream>>>(a, sequential);
                return MM_OK;
            });
            HIP_TRY(e, hipGetLastError(), "Error during calculate repeats / find lowest kernel");
        }
    } else if (n > 0) {
        const int blocks = sequential ? 1 : std::max(1, std::min(n, 1024));
        if ((st = ensure(e, e->scratch, (size_t)blocks * e->n_lib * sizeof(double), "Failed to allocate row scratch")))
            return st;
        a.scratch = (double*)e->scratch.p;
        st = by_type(diff_type, [&](auto T) {
            mm::resolve_kernel<decltype(T)::value><<<blocks, mm::kBlock, 0, e->stream>>>(a, sequential);
            return MM_OK;
        });
        HIP_TRY(e, hipGetLastError(), "Error during calculate repeats / find lowest kernel");
    }
    HIP_TRY(e, hipEventRecord(e->ev[5], e->stream), "event");

    // Copy best fit to host (CUDAPhotomosaicGenerator.cpp:305-306).
    static_assert(sizeof(long long) == sizeof(int64_t), "grid entries are 64-bit");
    HIP_TRY(e, hipMemcpyAsync(best_fit_out, e->grid.p, cells_total * sizeof(int64_t), hipMemcpyDeviceToHost, e->stream),
            "Error copying best fits to host");
    if (best_score_out)
        HIP_TRY(e, hipMemcpyAsync(best_score_out, e->best.p, cells_total * sizeof(double), hipMemcpyDeviceToHost, e->stream),
                "Error copying best scores to host");
    unsigned long long exact[2] = {0, 0};
    HIP_TRY(e, hipMemcpyAsync(exact, e->counter.p, sizeof exact, hipMemcpyDeviceToHost, e->stream), "counter");
    HIP_TRY(e, sync_stream(e), "Failed to synchronise stream");
    e->timing.resolve_ms += elapsed(e->ev[4], e->ev[5]);
    e->timing.exact_rescored += (int64_t)exact[0];
    e->timing.dense_rows += (int64_t)exact[1];
    return MM_OK;
}

// Page-locked host mirror of the result block.
mm_status ensure_pinned(mm_engine* e, size_t bytes)
{
    if (e->pinned && e->pinned_bytes >= bytes) return MM_OK;
    StageScope clock(&e->stage_clock.pinned_ms);
    if (e->pinned) { (void)hipHostFree(e->pinned); e->pinned = nullptr; e->pinned_bytes = 0; }
    HIP_TRY(e, hipHostMalloc(&e->pinned, bytes, hipHostMallocDefault), "Failed to allocate page-locked result buffer");
    e->pinned_bytes = bytes;
    return MM_OK;
}

// Single submission, part 1 (before any scoring): the result block -- header, grid, scores -- allocated and
// initialised on the device; the filter kernels count straight into the header from here on.
mm_status begin_single_submission(mm_engine* e, int n, int grid_w, int grid_h, bool speculate, bool keep_grid = false)
{
    // keep_grid: a later round of the same generate -- the block's grid is the fix point of the PREVIOUS lists and stays as the sweeps'
    // starting state (header and flags are cleared as ever)
    const size_t cells_total = (size_t)grid_w * grid_h;
    const size_t block_bytes = sizeof(mm::ResultHeader) + cells_total * (sizeof(long long) + sizeof(double));
    mm_status st;
    if ((st = ensure(e, e->result, block_bytes, "Failed to allocate result block")) ||
        (st = ensure_pinned(e, block_bytes)))
        return st;
    if (speculate &&
        ((st = ensure(e, e->deep, (size_t)n, "Failed to allocate list flags")) ||
         (st = ensure(e, e->insufficient, (size_t)n, "Failed to allocate list flags"))))
        return st;
    mm::ResultHeader* hdr = (mm::ResultHeader*)e->result.p;
    long long* grid_dev = (long long*)((char*)e->result.p + sizeof(mm::ResultHeader));
    double* best_dev = (double*)(grid_dev + cells_total);
    const size_t work = std::max(cells_total, speculate ? (size_t)n : 0);
    const unsigned fill_blocks = (unsigned)std::min<size_t>((work + mm::kBlock - 1) / mm::kBlock, 1024);
    mm::result_init_kernel<<<std::max(fill_blocks, 1u), mm::kBlock, 0, e->stream>>>(
        hdr, grid_dev, best_dev, keep_grid ? (size_t)0 : cells_total, (long long)e->n_lib,
        speculate ? (uint8_t*)e->deep.p : nullptr, speculate ? (uint8_t*)e->insufficient.p : nullptr, speculate ? n : 0);
    HIP_TRY(e, hipGetLastError(), "Error during result initialisation kernel");
    e->fast_hdr = hdr;
    return MM_OK;
}

// Single submission, part 2 (after the lists were enqueued): the resolve -- a handful of sweeps when the repeat
// penalty makes it order-dependent, ONE launch otherwise -- runs on the lists as they are, and header + grid
// (+ scores) come back in ONE copy into page-locked memory; the host waits once.  Whether the state in the block
// is final is for the caller to read off the header (all counters zero and a sweep that changed nothing).
// Needs the wave form of the resolve (the caller checks).  counted: the ranking kernel counted flags and marks
// into the header already; else two small kernels do it here.
mm_status resolve_single_submission(mm_engine* e, const mm_cell* recs_dev, int n, int grid_w, int grid_h,
                                    int diff_type, int range, int addition, int K, int k_fill, bool speculate,
                                    const int32_t* cand_i_dev, const double* cand_s_dev, bool want_scores,
                                    const uint8_t* flags_dev, bool counted, bool continuing = false, int sweeps_done = 0, bool warm = false)
{
    // warm: the block's grid is the previous round's fix point (begin_single_submission keep_grid): after a round of deepening only the
    // extended cells and what they displace change, so the sweeps -- hopeless from a cold grid on such content -- get a first set from
    // there before the front form is asked (a sweep that changes nothing proves the unique fix point whatever the start)
    // continuing: the block holds the state after sweeps that had not settled yet (nothing else wrong with it): the header is
    // cleared and another set of sweeps runs from that state -- a fix-point iteration may be resumed anywhere
    const size_t cells_total = (size_t)grid_w * grid_h;
    mm::ResultHeader* hdr = continuing ? (mm::ResultHeader*)e->result.p : e->fast_hdr;
    e->fast_hdr = nullptr;
    if (continuing) HIP_TRY(e, hipMemsetAsync(hdr, 0, sizeof(mm::ResultHeader), e->stream), "Failed to clear the result header");
    long long* grid_dev = (long long*)((char*)e->result.p + sizeof(mm::ResultHeader));
    double* best_dev = (double*)(grid_dev + cells_total);
    mm_status st;
    if (e->phase_events) HIP_TRY(e, hipEventRecord(e->ev[10], e->stream), "event");
    mm::ResolveArgs a = resolve_args_of(e, recs_dev, n, grid_w, grid_h, range, addition, K, cand_i_dev, cand_s_dev, grid_dev, best_dev,
                                        speculate ? (const uint8_t*)e->deep.p : nullptr, speculate ? (uint8_t*)e->insufficient.p : nullptr, k_fill);
    a.exact_count = &hdr->exact_plain;
    a.dense_count = &hdr->dense_count;
    a.row_slot = nullptr;                       // a mark voids the block; the step-by-step path then builds the rows
    a.insufficient_count = speculate ? &hdr->n_insufficient : nullptr;
    if (!counted && !continuing) {
        if ((st = ensure(e, e->row_slot, (size_t)n * sizeof(int) + sizeof(int), "Failed to allocate row slots"))) return st;
        mm::mark_scan_kernel<<<std::min((n + mm::kBlock - 1) / mm::kBlock, 256), mm::kBlock, 0, e->stream>>>(
            cand_i_dev, n, K, a.deep, (int*)e->row_slot.p, (int*)&hdr->n_marked);
        HIP_TRY(e, hipGetLastError(), "Error during mark scan kernel");
        if (flags_dev) {
            mm::result_summary_kernel<<<1, mm::kBlock, 0, e->stream>>>(hdr, flags_dev, n);
            HIP_TRY(e, hipGetLastError(), "Error during result summary kernel");
        }
    }
    const int sequential = (range > 0 && addition != 0) ? 1 : 0;
    const int blocks = std::max(1, std::min(n, 8192));
    int sweeps = 0;
    // The front form (kernels.hpp resolve_front_kernel: the exact wavefront in ONE launch, ~0.1 us per cell of its critical path) where the
    // sweeps do not settle in their first set: a continued block goes to it instead of further sets (photographs at config D's scale
    // had not settled after 240 sweeps), and once that has happened over a (library, cell size) the next generate starts with it.
    const size_t front_lds = wavefront_lds_bytes(grid_w, grid_h, true);
    const bool front = sequential && e->use_front_resolve != 0 && front_lds != 0 &&
                       (continuing || (!warm && e->front_learned && e->sweeps_lib == e->n_lib && e->sweeps_size == e->lib_size));
    e->result_grid_valid = false;
    e->last_front = front;
    if (front) {
        if (continuing) { e->front_learned = true; e->sweeps_learned = 0; e->sweeps_lib = e->n_lib; e->sweeps_size = e->lib_size; }
        a.sweep_changes = nullptr;
        a.prev_changes = nullptr;
        a.exact_count = &hdr->sweep_exact[0];
        a.insufficient_count = speculate ? &hdr->sweep_insufficient[0] : nullptr;
        // (the header was cleared: sweep_changes[0] stays 0 = "this pass left the grid at its fix point", which is what one exact pass does;
        // the kernel's stuck flag is the header's spare word)
        if ((st = launch_wavefront(e, a, recs_dev, n, grid_h, diff_type, true, front_lds, (int*)&hdr->pad))) return st;
        sweeps = 1;
    } else if (sequential) {
        const bool learned = e->sweeps_lib == e->n_lib && e->sweeps_size == e->lib_size && e->sweeps_learned > 8;
        const int want = continuing ? mm::kFastSweeps : (learned ? std::min(mm::kFastSweeps, (e->sweeps_learned + 2 + 7) / 8 * 8) : 8);
        // (what is left of "max_sweeps" bounds the set, not the option itself: a continued set used to enqueue its full want and the
        // sets of a call 8 + 32 + 32 = 72 sweeps under the default of 48 -- ADVICE round 5)
        sweeps = std::max(1, std::min(e->max_sweeps - sweeps_done, want));
        for (int i = 0; i < sweeps; ++i) {
            a.prev_changes = i > 0 ? &hdr->sweep_changes[i - 1] : nullptr;
            a.sweep_changes = &hdr->sweep_changes[i];
            a.exact_count = &hdr->sweep_exact[i];
            a.insufficient_count = speculate ? &hdr->sweep_insufficient[i] : nullptr;
            by_type(diff_type, [&](auto T) {
                mm::resolve_wave_kernel<decltype(T)::value><<<blocks, mm::kWave, 0, e->stream>>>(a, 2);
                return MM_OK;
            });
        }
    } else {
        by_type(diff_type, [&](auto T) {
            mm::resolve_wave_kernel<decltype(T)::value><<<blocks, mm::kWave, 0, e->stream>>>(a, 0);
            return MM_OK;
        });
    }
    HIP_TRY(e, hipGetLastError(), "Error during calculate repeats / find lowest kernel");
    if (e->phase_events) HIP_TRY(e, hipEventRecord(e->ev[11], e->stream), "event");
    const size_t back = sizeof(mm::ResultHeader) + cells_total * (sizeof(long long) + (want_scores ? sizeof(double) : 0));
    HIP_TRY(e, hipMemcpyAsync(e->pinned, e->result.p, back, hipMemcpyDeviceToHost, e->stream), "Error copying best fits to host");
    HIP_TRY(e, hipEventRecord(e->ev[7], e->stream), "event");
    HIP_TRY(e, sync_stream(e), "Failed to synchronise stream");
    mm::ResultHeader* h = (mm::ResultHeader*)e->pinned;
    h->sweeps = (unsigned int)sweeps;
    if (front && h->pad != 0u) { set_err(e, "wavefront resolve made no progress (internal error)"); return MM_DEVICE_ERROR; }
    if (sequential && sweeps > 0) {       // the final state's verdicts: those of the sweep that changed nothing (the ones after it did not run), else the last one's
        int last = sweeps - 1;
        for (int i = 0; i < sweeps; ++i)
            if (h->sweep_changes[i] == 0) { last = i; break; }
        h->n_insufficient = h->sweep_insufficient[last];
        e->result_grid_valid = h->sweep_changes[last] == 0u;     // (the front form leaves its one counter at 0)
    }
    h->edge_pairs = h->filter_counters[0];
    h->launch_void = h->filter_counters[mm::kFilterVoidWord];
    memcpy(h->work, h->filter_counters + 2, sizeof h->work);
    return MM_OK;
}

// The block's sweeps have not settled (content on which neighbours want the same few images -- photographs -- needs more than
// the eight enqueued at first): further sets of sweeps from the state the block holds, one copy and one wait each, until a
// sweep changes nothing or `max_sweeps` are spent.  Returns the index of the settling sweep within the LAST set (-1: none),
// and the sweeps run before that set in *before.
mm_status settle_sweeps(mm_engine* e, const mm_cell* recs_dev, int n, int grid_w, int grid_h, int diff_type, int range, int addition,
                        int K, int k_fill, bool speculate, const int32_t* cand_i_dev, const double* cand_s_dev, bool want_scores,
                        int* settled, int* before)
{
    const mm::ResultHeader& h = *(const mm::ResultHeader*)e->pinned;
    *before = 0;
    int done = (int)h.sweeps;
    while (*settled < 0 && done < e->max_sweeps) {
        if (e->phase_events) e->timing.resolve_ms += elapsed(e->ev[10], e->ev[11]);      // (the set that did not settle)
        mm_status st = resolve_single_submission(e, recs_dev, n, grid_w, grid_h, diff_type, range, addition, K, k_fill, speculate,
                                                 cand_i_dev, cand_s_dev, want_scores, nullptr, false, true, done);
        if (st) return st;
        *before = done;
        for (unsigned int i = 0; i < h.sweeps && *settled < 0; ++i)
            if (h.sweep_changes[i] == 0) *settled = (int)i;
        done += (int)h.sweeps;
    }
    return MM_OK;
}

// ---- resources kept across engines -----------------------------------------------------------------------------
// The reference builds a fresh generator per run (MainWindow.cpp:584-607) and its device path allocates and frees
// everything inside generateBestFits (CUDAPhotomosaicGenerator.cpp:82,356).  For its everyday job sizes that is most
// of what a call costs here too (mm_stage_timing: stream and event creation, ~40 hipMallocs, a page-locked buffer --
// milliseconds against a 0.1 ms generate).  So mm_destroy PARKS the engine's device resources -- stream, events,
// every device buffer, the page-locked result mirror -- instead of releasing them (one set per device), and the next
// mm_create on that device adopts them: buffers that are large enough are reused as they are (`ensure`), the two
// constant tables (hue table, sRGB decode table) stay valid, everything logical (what is staged, options, timings)
// starts from the defaults.  mm_trim_cache() releases what is parked; MOSAIC_HIP_KEEP=0 in the environment turns
// the parking off.
DevBuf mm_engine::* const kDeviceBuffers[] = {
    &mm_engine::lib, &mm_engine::masks, &mm_engine::target, &mm_engine::recs, &mm_engine::cells, &mm_engine::cmask, &mm_engine::scores,
    &mm_engine::cand_i, &mm_engine::cand_s, &mm_engine::grid, &mm_engine::best, &mm_engine::counter, &mm_engine::scratch,
    &mm_engine::cont_i, &mm_engine::cont_f, &mm_engine::cont_s, &mm_engine::flags, &mm_engine::edges, &mm_engine::edge_count,
    &mm_engine::dense_recs, &mm_engine::dense_i, &mm_engine::lib_means, &mm_engine::seeds, &mm_engine::seed_score, &mm_engine::limits,
    &mm_engine::row_slot, &mm_engine::dense_rows_buf, &mm_engine::row_index, &mm_engine::lib_chroma, &mm_engine::cell_chroma,
    &mm_engine::hue_lut, &mm_engine::row_map, &mm_engine::spec_recs, &mm_engine::deep, &mm_engine::insufficient, &mm_engine::lib_bgr,
    &mm_engine::owner, &mm_engine::mosaic, &mm_engine::lib_hwc, &mm_engine::perm, &mm_engine::inv_perm, &mm_engine::cell_var,
    &mm_engine::cell_full, &mm_engine::cell_order, &mm_engine::pos_mask, &mm_engine::lib_lsum, &mm_engine::cell_lsum,
    &mm_engine::stage_u8, &mm_engine::srgb_lut, &mm_engine::shrink_tmp, &mm_engine::covered, &mm_engine::cells_f0, &mm_engine::area_tab,
    &mm_engine::sweep_counts, &mm_engine::sentinel, &mm_engine::lib_il, &mm_engine::cells_il, &mm_engine::result, &mm_engine::main_bgr,
    &mm_engine::entropy_out, &mm_engine::exceeded_out, &mm_engine::lib_tiny, &mm_engine::lib_stats_t, &mm_engine::range_flags, &mm_engine::lib_h, &mm_engine::cells_h };

std::mutex g_park_lock;
std::vector<mm_engine*> g_parked;           // at most one per device: engines that only hold resources

bool parking_enabled()
{
    const char* v = std::getenv("MOSAIC_HIP_KEEP");
    return !(v && v[0] == '0');
}

// Device bytes a parked set may hold.  Default: a BUDGET AGAINST THE DEVICE'S FREE MEMORY -- 5 % of what would be free with the
// set released (hipMemGetInfo at mm_destroy: free + held), i.e. ~14 GB on an otherwise idle MI355X, so a config-D generator's 2.4 GB
// of workspaces stay and the next generator allocates nothing, while on a device other tenants have filled the set shrinks with what is
// left.  MOSAIC_HIP_KEEP_MB in the environment pins a fixed cap instead (0: only stream, events and the pinned mirror are kept).
// Round 5 shipped a fixed 256 MB on the measurement that config D's workspaces "allocate again in 0.3 ms"; on the driver's box the
// same hipMallocs took 73-76 ms (BENCH_r05.json first_call: 137 / 122 ms against 56 / 48 with everything parked) -- hipMalloc of a
// multi-GB buffer is not sub-millisecond on every box, and the reference's pattern (a generator per run, MainWindow.cpp:584-607) pays it
// every time.  What the cap is for -- memory a co-tenant (a second engine budgeting on hipMemGetInfo, PyTorch / RCCL in dist.py) does
// not get back until mm_trim_cache -- is bounded by the 5 %.
constexpr double kParkShareOfFree = 0.05;
size_t parked_bytes_cap(size_t held)
{
    const char* v = std::getenv("MOSAIC_HIP_KEEP_MB");
    if (v && v[0]) return (size_t)std::max<long long>(std::atoll(v), 0) << 20;
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) return (size_t)256 << 20;
    return (size_t)(kParkShareOfFree * (double)(free_b + held));
}

void free_resources(mm_engine* e)
{
    for (DevBuf mm_engine::* m : kDeviceBuffers) release(e->*m);
    if (e->pinned) (void)hipHostFree(e->pinned);
    e->pinned = nullptr; e->pinned_bytes = 0;
    for (auto& ev : e->ev) if (ev) { (void)hipEventDestroy(ev); ev = nullptr; }
    if (e->stream) { (void)hipStreamDestroy(e->stream); e->stream = nullptr; }
}

// Moves the resources of `from` into `to` (which holds none).
void move_resources(mm_engine* to, mm_engine* from)
{
    for (DevBuf mm_engine::* m : kDeviceBuffers) { to->*m = from->*m; from->*m = DevBuf{}; }
    to->pinned = from->pinned; to->pinned_bytes = from->pinned_bytes;
    from->pinned = nullptr; from->pinned_bytes = 0;
    to->stream = from->stream; from->stream = nullptr;
    for (int i = 0; i < 15; ++i) { to->ev[i] = from->ev[i]; from->ev[i] = nullptr; }
}

// mm_destroy: true if the engine's resources were parked (the caller deletes the shell).
bool park_resources(mm_engine* e)
{
    if (!parking_enabled() || !e->stream) return false;
    for (const auto& ev : e->ev) if (!ev) return false;       // (an engine whose creation failed half way)
    mm_engine* keep = new (std::nothrow) mm_engine();
    if (!keep) return false;
    std::lock_guard<std::mutex> hold(g_park_lock);
    for (const mm_engine* p : g_parked)
        if (p->device == e->device) { delete keep; return false; }       // one set per device: this one is released
    keep->device = e->device;
    move_resources(keep, e);
    // the largest buffers are released until the set fits its cap (the two constant tables are a few KB: they stay)
    size_t held = 0;
    for (DevBuf mm_engine::* m : kDeviceBuffers) held += (keep->*m).bytes;
    const size_t cap = parked_bytes_cap(held);
    while (held > cap) {
        DevBuf* largest = nullptr;
        for (DevBuf mm_engine::* m : kDeviceBuffers)
            if ((keep->*m).p && (!largest || (keep->*m).bytes > largest->bytes)) largest = &(keep->*m);
        if (!largest) break;
        held -= largest->bytes;
        release(*largest);
    }
    g_parked.push_back(keep);
    return true;
}

// mm_create: the parked resources of this device, if any.
bool adopt_resources(mm_engine* e)
{
    mm_engine* from = nullptr;
    {
        std::lock_guard<std::mutex> hold(g_park_lock);
        for (size_t i = 0; i < g_parked.size(); ++i)
            if (g_parked[i]->device == e->device) { from = g_parked[i]; g_parked.erase(g_parked.begin() + (long)i); break; }
    }
    if (!from) return false;
    move_resources(e, from);
    delete from;
    return true;
}

int64_t count_pair_evals(const mm_engine* e, const mm_cell* cells, int n)
{
    // Upper bound used for reporting: bound area per cell x candidates (the mask
    // thins it further; bench.py computes the exact active-pixel figure itself).
    int64_t px = 0;
    for (int i = 0; i < n; ++i)
        px += (int64_t)(cells[i].row_max - cells[i].row_min) * (cells[i].col_max - cells[i].col_min);
    return px * e->n_lib;
}

}  // namespace

// ===========================================================================
// C ABI
// ===========================================================================

extern "C" {

int mm_abi_version(void) { return MM_ABI_VERSION; }

int mm_debug_has_pretier(void) { return MM_WITH_PRETIER != 0 ? 1 : 0; }

mm_status mm_device_count(int* count)
{
    if (!count) return MM_INVALID_ARGUMENT;
    *count = 0;
    hipError_t code = hipGetDeviceCount(count);
    if (code != hipSuccess) {
        *count = 0;
        return hip_fail(nullptr, code, "Failed to count devices", __FILE__, __LINE__);
    }
    return MM_OK;
}

mm_status mm_create(int device, mm_engine** out)
{
    if (!out) return MM_INVALID_ARGUMENT;
    *out = nullptr;
    int count = 0;
    hipError_t code = hipGetDeviceCount(&count);
    if (code != hipSuccess) return hip_fail(nullptr, code, "Failed to count devices", __FILE__, __LINE__);
    if (device < 0 || device >= count) {
        set_err(nullptr, "Invalid device %d (%d visible)", device, count);
        return MM_INVALID_ARGUMENT;
    }
    const double t_create = wall_ms();
    mm_engine* e = new (std::nothrow) mm_engine();
    if (!e) { set_err(nullptr, "out of host memory"); return MM_OUT_OF_MEMORY; }
    e->device = device;
    code = hipSetDevice(device);
    if (code == hipSuccess && adopt_resources(e)) {
        e->stage_clock.reused_engine = 1;
    } else {
        if (code == hipSuccess) code = hipStreamCreateWithFlags(&e->stream, hipStreamNonBlocking);
        for (int i = 0; code == hipSuccess && i < 15; ++i) code = hipEventCreate(&e->ev[i]);
    }
    if (code != hipSuccess) {
        mm_status st = hip_fail(nullptr, code, "Failed to initialise device", __FILE__, __LINE__);
        mm_destroy(e);
        return st;
    }
    e->stage_clock.create_ms = wall_ms() - t_create;
    *out = e;
    return MM_OK;
}

mm_status mm_get_stage_timing(const mm_engine* e, mm_stage_timing* out)
{
    if (!e || !out) return MM_INVALID_ARGUMENT;
    *out = e->stage_clock;
    return MM_OK;
}

void mm_destroy(mm_engine* e)
{
    if (!e) return;
    (void)hipSetDevice(e->device);
    if (e->stream) (void)sync_stream(e);
    if (park_resources(e)) { delete e; return; }       // kept for the next mm_create on this device (mm_trim_cache frees them)
    free_resources(e);
    delete e;
}

int mm_trim_cache(void)
{
    std::vector<mm_engine*> parked;
    {
        std::lock_guard<std::mutex> hold(g_park_lock);
        parked.swap(g_parked);
    }
    for (mm_engine* p : parked) {
        (void)hipSetDevice(p->device);
        free_resources(p);
        delete p;
    }
    return (int)parked.size();
}

const char* mm_last_error(const mm_engine* e) { return e ? e->err.c_str() : g_create_error.c_str(); }

mm_status mm_set_library(mm_engine* e, const float* images, int n_images, int size)
{
    if (!e) return MM_INVALID_ARGUMENT;
    if (!images || n_images <= 0 || size <= 0 || size > 4096) {
        set_err(e, "mm_set_library: need n_images > 0 and 0 < size <= 4096 (got %d, %d)", n_images, size);
        return MM_INVALID_ARGUMENT;
    }
    HIP_TRY(e, select_device(e), "Failed to select device");
    const size_t bytes = (size_t)n_images * size * size * 3 * sizeof(float);
    mm_status st = ensure(e, e->lib_hwc, bytes, "Failed to allocate library images");
    if (st) return st;
    {
        StageScope clock(&e->stage_clock.h2d_ms);
        HIP_TRY(e, hipMemcpy(e->lib_hwc.p, images, bytes, hipMemcpyHostToDevice), "Failed to copy library image to device");
    }
    e->planes_dirty = true;        // planes are (re)built by prepare_planes() once the masks are known
    e->n_lib = n_images;
    e->lib_size = size;
    return value_range(e, (const float*)e->lib_hwc.p, (size_t)n_images * size * size, &e->lib_range);
}

mm_status mm_set_masks(mm_engine* e, const uint8_t* masks, int size)
{
    if (!e) return MM_INVALID_ARGUMENT;
    if (!masks || size <= 0 || size > 4096) {
        set_err(e, "mm_set_masks: need 0 < size <= 4096 (got %d)", size);
        return MM_INVALID_ARGUMENT;
    }
    HIP_TRY(e, select_device(e), "Failed to select device");
    const int P = size * size, ppad = round_up(P, 256);
    std::vector<uint8_t> padded((size_t)4 * ppad, 0);
    for (int m = 0; m < 4; ++m) memcpy(padded.data() + (size_t)m * ppad, masks + (size_t)m * P, (size_t)P);
    mm_status st = ensure(e, e->masks, padded.size(), "Failed to allocate cell masks");
    if (st) return st;
    {
        StageScope clock(&e->stage_clock.h2d_ms);
        HIP_TRY(e, hipMemcpy(e->masks.p, padded.data(), padded.size(), hipMemcpyHostToDevice),
                "Failed to copy cell mask to device");
    }
    e->masks_host.assign(masks, masks + (size_t)4 * P);
    e->planes_dirty = true;
    e->mask_size = size;
    return MM_OK;
}

mm_status mm_set_target(mm_engine* e, const float* image, int rows, int cols)
{
    if (!e) return MM_INVALID_ARGUMENT;
    if (!image || rows <= 0 || cols <= 0) {
        set_err(e, "mm_set_target: need rows, cols > 0 (got %d, %d)", rows, cols);
        return MM_INVALID_ARGUMENT;
    }
    HIP_TRY(e, select_device(e), "Failed to select device");
    const size_t bytes = (size_t)rows * cols * 3 * sizeof(float);
    mm_status st = ensure(e, e->target, bytes, "Failed to allocate main image");
    if (st) return st;
    {
        StageScope clock(&e->stage_clock.h2d_ms);
        HIP_TRY(e, hipMemcpy(e->target.p, image, bytes, hipMemcpyHostToDevice), "Failed to copy main image to device");
    }
    e->rows = rows;
    e->cols = cols;
    e->pre_auto = mm_engine::PreAuto{};          // other content: the pre-tier is measured again
    return value_range(e, (const float*)e->target.p, (size_t)rows * cols, &e->target_range);
}

namespace {

// sRGB decode table for preprocess_u8_kernel: c/255 -> linear, fp64 (the formulae of
// synth.bgr_u8_to_lab_f32).
mm_status ensure_srgb_lut(mm_engine* e)
{
    if (e->srgb_lut.p) return MM_OK;
    double lut[256];
    for (int i = 0; i < 256; ++i) {
        const double c = (double)i / 255.0;
        lut[i] = c > 0.04045 ? std::pow((c + 0.055) / 1.055, 2.4) : c / 12.92;
    }
    mm_status st = ensure(e, e->srgb_lut, sizeof lut, "Failed to allocate sRGB table");
    if (st) return st;
    const hipError_t code = hipMemcpy(e->srgb_lut.p, lut, sizeof lut, hipMemcpyHostToDevice);     // (non-null = filled, as the hue table)
    if (code != hipSuccess) { release(e->srgb_lut); return hip_fail(e, code, "Failed to copy sRGB table to device", __FILE__, __LINE__); }
    return MM_OK;
}

// u8 BGR images -> resident f32 (dst), optionally shrunk by `factor` first.
// computeResizeAreaTab (OpenCV 4 imgproc resize.cpp) for one axis: the source indices and
// weights under each destination index, scale = 1 / (dsize / ssize) as cv::resize computes it.
void build_area_table(int ssize, int dsize, std::vector<int>& first, std::vector<int>& src, std::vector<float>& alpha)
{
    const double scale = 1.0 / ((double)dsize / ssize);
    first.assign(dsize + 1, 0);
    src.clear(); alpha.clear();
    for (int dx = 0; dx < dsize; ++dx) {
        const double fsx1 = dx * scale, fsx2 = fsx1 + scale;
        const double cell = std::min(scale, ssize - fsx1);
        int sx1 = (int)std::ceil(fsx1), sx2 = (int)std::floor(fsx2);
        sx2 = std::min(sx2, ssize - 1);
        sx1 = std::min(sx1, sx2);
        first[dx] = (int)src.size();
        if (sx1 - fsx1 > 1e-3) { src.push_back(sx1 - 1); alpha.push_back((float)((sx1 - fsx1) / cell)); }
        for (int sx = sx1; sx < sx2; ++sx) { src.push_back(sx); alpha.push_back((float)(1.0 / cell)); }
        if (fsx2 - sx2 > 1e-3) { src.push_back(sx2); alpha.push_back((float)(std::min(std::min(fsx2 - sx2, 1.0), cell) / cell)); }
    }
    first[dsize] = (int)src.size();
}

mm_status preprocess_u8(mm_engine* e, const uint8_t* bgr, size_t n_images, int rows, int cols, int factor,
                        int diff_type, DevBuf& dst)
{
    const size_t in_bytes = n_images * (size_t)rows * cols * 3;
    const size_t out_px = n_images * (size_t)(rows / factor) * (cols / factor);
    mm_status st;
    if ((st = ensure_srgb_lut(e)) || (st = ensure(e, e->stage_u8, in_bytes, "Failed to allocate u8 staging")) ||
        (st = ensure(e, dst, out_px * 3 * sizeof(float), "Failed to allocate preprocessed images")))
        return st;
    HIP_TRY(e, hipEventRecord(e->ev[3], e->stream), "event");
    HIP_TRY(e, hipMemcpyAsync(e->stage_u8.p, bgr, in_bytes, hipMemcpyHostToDevice, e->stream), "Failed to copy u8 images to device");
    const unsigned blocks = (unsigned)std::min<size_t>((out_px + mm::kBlock - 1) / mm::kBlock, 65536);
    HIP_TRY(e, hipEventRecord(e->ev[4], e->stream), "event");
    mm::preprocess_u8_kernel<<<std::max(blocks, 1u), mm::kBlock, 0, e->stream>>>(
        (const uint8_t*)e->stage_u8.p, (float*)dst.p, n_images, rows, cols, factor,
        diff_type != MM_RGB_EUCLIDEAN, (const double*)e->srgb_lut.p);
    HIP_TRY(e, hipGetLastError(), "Error during preprocessing kernel");
    HIP_TRY(e, hipEventRecord(e->ev[5], e->stream), "event");
    HIP_TRY(e, sync_stream(e), "Failed to synchronise stream");
    e->timing = mm_timing{};
    e->timing.preprocess_ms = elapsed(e->ev[4], e->ev[5]);
    e->stage_clock.h2d_ms += elapsed(e->ev[3], e->ev[4]);
    e->stage_clock.preprocess_ms += e->timing.preprocess_ms;
    return MM_OK;
}

}  // namespace

mm_status mm_set_library_u8(mm_engine* e, const uint8_t* bgr, int n_images, int size, int shrink, int diff_type)
{
    if (!e) return MM_INVALID_ARGUMENT;
    if (!bgr || n_images <= 0 || size <= 0 || size > 4096 || shrink < 1 || size % shrink != 0 || !known_type(diff_type)) {
        set_err(e, "mm_set_library_u8: need n_images > 0, 0 < size <= 4096, shrink dividing size, a known type "
                   "(got %d, %d, %d, %d)", n_images, size, shrink, diff_type);
        return MM_INVALID_ARGUMENT;
    }
    HIP_TRY(e, select_device(e), "Failed to select device");
    mm_status st = preprocess_u8(e, bgr, (size_t)n_images, size, size, shrink, diff_type, e->lib_hwc);
    if (st) return st;
    e->planes_dirty = true;
    e->n_lib = n_images;
    e->lib_size = size / shrink;
    e->lib_range = 0;              // converted here: inside the box by construction
    return MM_OK;
}

mm_status mm_set_library_u8_sized(mm_engine* e, const uint8_t* bgr, int n_images, int size, int detail_size, int diff_type)
{
    if (!e) return MM_INVALID_ARGUMENT;
    if (!bgr || n_images <= 0 || size <= 0 || size > 4096 || detail_size < 1 || detail_size > size || !known_type(diff_type)) {
        set_err(e, "mm_set_library_u8_sized: need n_images > 0, 0 < detail size <= size <= 4096, a known type "
                   "(got %d, %d, %d, %d)", n_images, size, detail_size, diff_type);
        return MM_INVALID_ARGUMENT;
    }
    if (size % detail_size == 0) return mm_set_library_u8(e, bgr, n_images, size, size / detail_size, diff_type);
    HIP_TRY(e, select_device(e), "Failed to select device");
    // INTER_AREA by a ratio that is not whole, on the u8 images (the reference resizes before it
    // converts, PhotomosaicGeneratorBase.cpp:262-271), then the usual colour conversion
    std::vector<int> first, src;
    std::vector<float> alpha;
    build_area_table(size, detail_size, first, src, alpha);
    const size_t n = src.size(), in_bytes = (size_t)n_images * size * size * 3;
    const size_t out_px = (size_t)n_images * detail_size * detail_size;
    DevBuf tab, small;
    mm_status st;
    if ((st = ensure_srgb_lut(e)) || (st = ensure(e, e->stage_u8, in_bytes, "Failed to allocate u8 staging")) ||
        (st = ensure(e, tab, ((size_t)detail_size + 1 + 2 * n) * sizeof(int), "Failed to allocate resize table")) ||
        (st = ensure(e, small, out_px * 3, "Failed to allocate resized images")) ||
        (st = ensure(e, e->lib_hwc, out_px * 3 * sizeof(float), "Failed to allocate preprocessed images"))) {
        release(tab); release(small);
        return st;
    }
    int* base = (int*)tab.p;
    hipError_t code = hipMemcpyAsync(e->stage_u8.p, bgr, in_bytes, hipMemcpyHostToDevice, e->stream);
    if (code == hipSuccess) code = hipMemcpyAsync(base, first.data(), first.size() * sizeof(int), hipMemcpyHostToDevice, e->stream);
    if (code == hipSuccess) code = hipMemcpyAsync(base + detail_size + 1, src.data(), n * sizeof(int), hipMemcpyHostToDevice, e->stream);
    if (code == hipSuccess) code = hipMemcpyAsync(base + detail_size + 1 + n, alpha.data(), n * sizeof(float), hipMemcpyHostToDevice, e->stream);
    if (code == hipSuccess) code = hipEventRecord(e->ev[4], e->stream);
    if (code == hipSuccess) {
        const mm::CellResize rs{0, base, base + detail_size + 1, (const float*)(base + detail_size + 1 + n)};
        const unsigned blocks = (unsigned)std::min<size_t>((out_px + mm::kBlock - 1) / mm::kBlock, 65536);
        mm::resize_area_u8_kernel<<<std::max(blocks, 1u), mm::kBlock, 0, e->stream>>>(
            (const uint8_t*)e->stage_u8.p, (uint8_t*)small.p, (size_t)n_images, size, detail_size, rs);
        code = hipGetLastError();
        if (code == hipSuccess) {
            mm::preprocess_u8_kernel<<<std::max(blocks, 1u), mm::kBlock, 0, e->stream>>>(
                (const uint8_t*)small.p, (float*)e->lib_hwc.p, (size_t)n_images, detail_size, detail_size, 1,
                diff_type != MM_RGB_EUCLIDEAN, (const double*)e->srgb_lut.p);
            code = hipGetLastError();
        }
    }
    if (code == hipSuccess) code = hipEventRecord(e->ev[5], e->stream);
    if (code == hipSuccess) code = sync_stream(e);      // host vectors and temporaries go out of scope
    release(tab); release(small);
    if (code != hipSuccess) return hip_fail(e, code, "Error during library resize", __FILE__, __LINE__);
    e->timing = mm_timing{};
    e->timing.preprocess_ms = elapsed(e->ev[4], e->ev[5]);
    e->planes_dirty = true;
    e->n_lib = n_images;
    e->lib_size = detail_size;
    e->lib_range = 0;
    return MM_OK;
}

mm_status mm_set_target_u8(mm_engine* e, const uint8_t* bgr, int rows, int cols, int diff_type)
{
    if (!e) return MM_INVALID_ARGUMENT;
    if (!bgr || rows <= 0 || cols <= 0 || !known_type(diff_type)) {
        set_err(e, "mm_set_target_u8: need rows, cols > 0 and a known type (got %d, %d, %d)", rows, cols, diff_type);
        return MM_INVALID_ARGUMENT;
    }
    HIP_TRY(e, select_device(e), "Failed to select device");
    mm_status st = preprocess_u8(e, bgr, 1, rows, cols, 1, diff_type, e->target);
    if (st) return st;
    e->rows = rows;
    e->cols = cols;
    e->target_range = 0;
    e->pre_auto = mm_engine::PreAuto{};
    return MM_OK;
}

mm_status mm_hue_rotate_u8(mm_engine* e, const uint8_t* bgr, int rows, int cols, float degrees, uint8_t* out)
{
    if (!e) return MM_INVALID_ARGUMENT;
    if (!bgr || !out || rows <= 0 || cols <= 0 || !(degrees >= 0.0f && degrees < 360.0f)) {
        set_err(e, "mm_hue_rotate_u8: need rows, cols > 0 and 0 <= degrees < 360 (got %d, %d, %g)", rows, cols, (double)degrees);
        return MM_INVALID_ARGUMENT;
    }
    HIP_TRY(e, select_device(e), "Failed to select device");
    const size_t n_px = (size_t)rows * cols, bytes = n_px * 3;
    mm_status st;
    if ((st = ensure(e, e->stage_u8, 2 * bytes, "Failed to allocate u8 staging"))) return st;
    uint8_t* src = (uint8_t*)e->stage_u8.p;
    uint8_t* dst = src + bytes;
    HIP_TRY(e, hipMemcpyAsync(src, bgr, bytes, hipMemcpyHostToDevice, e->stream), "Failed to copy u8 image to device");
    const unsigned blocks = (unsigned)std::min<size_t>((n_px + mm::kBlock - 1) / mm::kBlock, 65536);
    mm::hue_rotate_u8_kernel<<<std::max(blocks, 1u), mm::kBlock, 0, e->stream>>>(src, dst, n_px, degrees);
    HIP_TRY(e, hipGetLastError(), "Error during hue rotation kernel");
    HIP_TRY(e, hipMemcpyAsync(out, dst, bytes, hipMemcpyDeviceToHost, e->stream), "Failed to copy variant to host");
    HIP_TRY(e, sync_stream(e), "Failed to synchronise stream");
    return MM_OK;
}

mm_status mm_shrink_library(mm_engine* e, int factor)
{
    if (!e) return MM_INVALID_ARGUMENT;
    if (!e->lib_hwc.p || e->n_lib <= 0) { set_err(e, "mm_shrink_library: no library staged"); return MM_NOT_READY; }
    if (factor < 1 || e->lib_size % factor != 0) {
        set_err(e, "mm_shrink_library: factor %d does not divide the image size %d (only integer "
                   "INTER_AREA factors are supported)", factor, e->lib_size);
        return MM_INVALID_ARGUMENT;
    }
    if (factor == 1) return MM_OK;
    HIP_TRY(e, select_device(e), "Failed to select device");
    const int so = e->lib_size / factor;
    const size_t out_px = (size_t)e->n_lib * so * so;
    mm_status st = ensure(e, e->shrink_tmp, out_px * 3 * sizeof(float), "Failed to allocate library images");
    if (st) return st;
    const unsigned blocks = (unsigned)std::min<size_t>((out_px + mm::kBlock - 1) / mm::kBlock, 65536);
    mm::shrink_f32_kernel<<<std::max(blocks, 1u), mm::kBlock, 0, e->stream>>>(
        (const float*)e->lib_hwc.p, (float*)e->shrink_tmp.p, e->n_lib, e->lib_size, factor);
    HIP_TRY(e, hipGetLastError(), "Error during library resize kernel");
    HIP_TRY(e, sync_stream(e), "Failed to synchronise stream");
    std::swap(e->lib_hwc, e->shrink_tmp);
    e->lib_size = so;
    e->planes_dirty = true;
    return MM_OK;
}

mm_status mm_set_cell_resize(mm_engine* e, int normal_size, int detail_size)
{
    if (!e) return MM_INVALID_ARGUMENT;
    if (detail_size < 1 || normal_size < detail_size || normal_size > 64 * detail_size || normal_size > 8192) {
        set_err(e, "mm_set_cell_resize: need 1 <= detail size <= normal size <= 64 x detail size (got %d, %d)", normal_size, detail_size);
        return MM_INVALID_ARGUMENT;
    }
    if (normal_size % detail_size == 0) { e->cell_scale = normal_size / detail_size; return MM_OK; }
    if (detail_size != e->lib_size) {
        set_err(e, "mm_set_cell_resize: detail size %d differs from the library's %d", detail_size, e->lib_size);
        return MM_INVALID_ARGUMENT;
    }
    HIP_TRY(e, select_device(e), "Failed to select device");
    std::vector<int> first, src;
    std::vector<float> alpha;
    build_area_table(normal_size, detail_size, first, src, alpha);
    const size_t n = src.size();
    mm_status st = ensure(e, e->area_tab, ((size_t)detail_size + 1 + 2 * n) * sizeof(int), "Failed to allocate resize table");
    if (st) return st;
    int* base = (int*)e->area_tab.p;
    HIP_TRY(e, hipMemcpy(base, first.data(), first.size() * sizeof(int), hipMemcpyHostToDevice), "Failed to copy resize table");
    HIP_TRY(e, hipMemcpy(base + detail_size + 1, src.data(), n * sizeof(int), hipMemcpyHostToDevice), "Failed to copy resize table");
    HIP_TRY(e, hipMemcpy(base + detail_size + 1 + n, alpha.data(), n * sizeof(float), hipMemcpyHostToDevice), "Failed to copy resize table");
    e->area_entries = (int)n;
    e->cell_normal = normal_size;
    e->area_detail = detail_size;
    e->cell_scale = 0;
    return MM_OK;
}

mm_status mm_set_cell_scale(mm_engine* e, int factor)
{
    if (!e) return MM_INVALID_ARGUMENT;
    if (factor < 1 || factor > 64) { set_err(e, "mm_set_cell_scale: need 1 <= factor <= 64 (got %d)", factor); return MM_INVALID_ARGUMENT; }
    e->cell_scale = factor;
    return MM_OK;
}

mm_status mm_get_library(mm_engine* e, float* out)
{
    if (!e || !out) return MM_INVALID_ARGUMENT;
    if (!e->lib_hwc.p || e->n_lib <= 0) { set_err(e, "mm_get_library: no library staged"); return MM_NOT_READY; }
    HIP_TRY(e, select_device(e), "Failed to select device");
    HIP_TRY(e, hipMemcpy(out, e->lib_hwc.p, (size_t)e->n_lib * e->lib_size * e->lib_size * 3 * sizeof(float),
                         hipMemcpyDeviceToHost), "Error copying library to host");
    return MM_OK;
}

mm_status mm_get_target(mm_engine* e, float* out)
{
    if (!e || !out) return MM_INVALID_ARGUMENT;
    if (!e->target.p || e->rows <= 0) { set_err(e, "mm_get_target: no main image staged"); return MM_NOT_READY; }
    HIP_TRY(e, select_device(e), "Failed to select device");
    HIP_TRY(e, hipMemcpy(out, e->target.p, (size_t)e->rows * e->cols * 3 * sizeof(float), hipMemcpyDeviceToHost),
            "Error copying main image to host");
    return MM_OK;
}

mm_status mm_library_shape(mm_engine* e, int* n_images, int* size)
{
    if (!e) return MM_INVALID_ARGUMENT;
    if (n_images) *n_images = e->n_lib;
    if (size) *size = e->lib_size;
    return MM_OK;
}

int mm_candidates_per_cell(int n_images, int repeat_range, int repeat_addition)
{
    if (n_images <= 0) return 0;
    long long k = 2;     // winner + runner-up (near-tie check)
    if (repeat_range > 0 && repeat_addition > 0) {
        // At most R(2R+1)+R neighbours can be penalised; a positive addition can
        // only push candidates back, so the winner is among the R(2R+1)+R+1 lowest.
        const long long r = repeat_range;
        k = r * (2 * r + 1) + r + 2;
    } else if (repeat_range > 0 && repeat_addition < 0) {
        k = n_images;    // a negative "penalty" can promote any candidate
    }
    return (int)std::min<long long>(k, n_images);
}

int mm_speculative_candidates(int n_images, int repeat_range, int repeat_addition, int spec_k)
{
    const int k = mm_candidates_per_cell(n_images, repeat_range, repeat_addition);
    // Short lists pay off only when the full bound is long and penalties can only push back.
    if (spec_k <= 0 || repeat_range <= 0 || repeat_addition <= 0 || k <= spec_k + 2 || spec_k > 64) return k;
    return spec_k;
}

mm_status mm_score_cells_partial(mm_engine* e, const mm_cell* cells, int n_cells, int diff_type, int k, int k_fill,
                                 int32_t* cand_index_out, double* cand_score_out, int outputs_on_device,
                                 mm_progress_cb progress, void* progress_user, volatile int* cancel)
{
    if (!e) return MM_INVALID_ARGUMENT;
    if (!known_type(diff_type)) { set_err(e, "No function for given type %d", diff_type); return MM_INVALID_ARGUMENT; }
    if (n_cells < 0 || (n_cells > 0 && !cells) || k <= 0 || k_fill <= 0 || k_fill > k || !cand_index_out || !cand_score_out) {
        set_err(e, "mm_score_cells: bad arguments");
        return MM_INVALID_ARGUMENT;
    }
    mm_status st = check_ready(e);
    if (st) return st;
    if ((st = check_cells(e, cells, n_cells, 0, 0))) return st;
    HIP_TRY(e, select_device(e), "Failed to select device");
    e->timing = mm_timing{};
    e->syncs = 0;
    e->phase_events = e->phase_timing != 0;
    e->fast_hdr = nullptr;
    if (n_cells == 0) return MM_OK;

    HIP_TRY(e, hipEventRecord(e->ev[6], e->stream), "event");
    if ((st = ensure(e, e->recs, (size_t)n_cells * sizeof(mm_cell), "Failed to allocate cell records"))) return st;
    HIP_TRY(e, hipMemcpyAsync(e->recs.p, cells, (size_t)n_cells * sizeof(mm_cell), hipMemcpyHostToDevice, e->stream),
            "Failed to copy cell records to device");
    int32_t* ci = cand_index_out;
    double* cs = cand_score_out;
    if (!outputs_on_device) {
        if ((st = ensure(e, e->cand_i, (size_t)n_cells * k * sizeof(int32_t), "Failed to allocate candidates")) ||
            (st = ensure(e, e->cand_s, (size_t)n_cells * k * sizeof(double), "Failed to allocate candidates")))
            return st;
        ci = (int32_t*)e->cand_i.p;
        cs = (double*)e->cand_s.p;
    }
    st = score_cells_device(e, cells, (const mm_cell*)e->recs.p, n_cells, diff_type, k_fill, k, ci, cs, nullptr,
                            progress, progress_user, cancel);
    if (st) return st;
    if (!outputs_on_device) {
        HIP_TRY(e, hipMemcpyAsync(cand_index_out, ci, (size_t)n_cells * k * sizeof(int32_t), hipMemcpyDeviceToHost, e->stream),
                "Error copying candidates to host");
        HIP_TRY(e, hipMemcpyAsync(cand_score_out, cs, (size_t)n_cells * k * sizeof(double), hipMemcpyDeviceToHost, e->stream),
                "Error copying candidates to host");
    }
    HIP_TRY(e, hipEventRecord(e->ev[7], e->stream), "event");
    HIP_TRY(e, sync_stream(e), "Failed to synchronise stream");
    e->timing.total_ms = elapsed(e->ev[6], e->ev[7]);
    e->timing.pair_evals = count_pair_evals(e, cells, n_cells);
    e->timing.host_syncs = e->syncs;
    return MM_OK;
}

mm_status mm_score_cells(mm_engine* e, const mm_cell* cells, int n_cells, int diff_type, int k,
                         int32_t* cand_index_out, double* cand_score_out, int outputs_on_device,
                         mm_progress_cb progress, void* progress_user, volatile int* cancel)
{
    return mm_score_cells_partial(e, cells, n_cells, diff_type, k, k, cand_index_out, cand_score_out,
                                  outputs_on_device, progress, progress_user, cancel);
}

mm_status mm_resolve_verify(mm_engine* e, const mm_cell* cells, int n_cells, int grid_w, int grid_h, int diff_type,
                            int repeat_range, int repeat_addition, int k, int k_fill,
                            const int32_t* cand_index, const double* cand_score, int inputs_on_device,
                            const uint8_t* deep, uint8_t* insufficient_out,
                            int64_t* best_fit_out, double* best_score_out)
{
    if (!e) return MM_INVALID_ARGUMENT;
    if (!known_type(diff_type)) { set_err(e, "No function for given type %d", diff_type); return MM_INVALID_ARGUMENT; }
    if (n_cells < 0 || (n_cells > 0 && (!cells || !cand_index || !cand_score)) || k <= 0 || grid_w <= 0 ||
        grid_h <= 0 || !best_fit_out || repeat_range < 0 || k_fill <= 0 || k_fill > k || (deep && !insufficient_out)) {
        set_err(e, "mm_resolve: bad arguments");
        return MM_INVALID_ARGUMENT;
    }
    mm_status st = check_ready(e);
    if (st) return st;
    if ((st = check_cells(e, cells, n_cells, grid_w, grid_h))) return st;
    HIP_TRY(e, select_device(e), "Failed to select device");
    if ((st = ensure(e, e->recs, (size_t)std::max(n_cells, 1) * sizeof(mm_cell), "Failed to allocate cell records"))) return st;
    HIP_TRY(e, hipMemcpyAsync(e->recs.p, cells, (size_t)n_cells * sizeof(mm_cell), hipMemcpyHostToDevice, e->stream),
            "Failed to copy cell records to device");
    const int32_t* ci = cand_index;
    const double* cs = cand_score;
    if (!inputs_on_device && n_cells > 0) {
        if ((st = ensure(e, e->cand_i, (size_t)n_cells * k * sizeof(int32_t), "Failed to allocate candidates")) ||
            (st = ensure(e, e->cand_s, (size_t)n_cells * k * sizeof(double), "Failed to allocate candidates")))
            return st;
        HIP_TRY(e, hipMemcpyAsync(e->cand_i.p, cand_index, (size_t)n_cells * k * sizeof(int32_t), hipMemcpyHostToDevice, e->stream),
                "Failed to copy candidates to device");
        HIP_TRY(e, hipMemcpyAsync(e->cand_s.p, cand_score, (size_t)n_cells * k * sizeof(double), hipMemcpyHostToDevice, e->stream),
                "Failed to copy candidates to device");
        ci = (const int32_t*)e->cand_i.p;
        cs = (const double*)e->cand_s.p;
    }
    const uint8_t* deep_dev = nullptr;
    uint8_t* insuff_dev = nullptr;
    if (deep && n_cells > 0) {
        if ((st = ensure(e, e->deep, (size_t)n_cells, "Failed to allocate list flags")) ||
            (st = ensure(e, e->insufficient, (size_t)n_cells, "Failed to allocate list flags")))
            return st;
        HIP_TRY(e, hipMemcpyAsync(e->deep.p, deep, (size_t)n_cells, hipMemcpyHostToDevice, e->stream), "Failed to copy list flags");
        HIP_TRY(e, hipMemsetAsync(e->insufficient.p, 0, (size_t)n_cells, e->stream), "Failed to clear list flags");
        deep_dev = (const uint8_t*)e->deep.p;
        insuff_dev = (uint8_t*)e->insufficient.p;
    }
    st = resolve_device(e, (const mm_cell*)e->recs.p, n_cells, grid_w, grid_h, diff_type, repeat_range,
                        repeat_addition, k, ci, cs, best_fit_out, best_score_out, deep_dev, insuff_dev, k_fill);
    if (st) return st;
    if (insuff_dev)
        HIP_TRY(e, hipMemcpy(insufficient_out, insuff_dev, (size_t)n_cells, hipMemcpyDeviceToHost), "Error copying list flags to host");
    return MM_OK;
}

mm_status mm_resolve(mm_engine* e, const mm_cell* cells, int n_cells, int grid_w, int grid_h, int diff_type,
                     int repeat_range, int repeat_addition, int k,
                     const int32_t* cand_index, const double* cand_score, int inputs_on_device,
                     int64_t* best_fit_out, double* best_score_out)
{
    return mm_resolve_verify(e, cells, n_cells, grid_w, grid_h, diff_type, repeat_range, repeat_addition, k, k,
                             cand_index, cand_score, inputs_on_device, nullptr, nullptr, best_fit_out, best_score_out);
}

// generateBestFits.  With a repeat penalty the counting bound asks for K = R(2R+1)+R+2
// candidates per cell, which is almost never needed: score short lists of K_s entries
// first ("speculate"), let the resolve verify every pick on the device (an unlisted
// candidate has a base score >= the list's last one and a penalty >= 0), then deepen the
// few cells it could not vouch for to the full K and resolve again, until none is left.
// Deep lists are sufficient by the counting argument, so the loop ends, and the result is
// the one the full-K path gives.
//
// Single submission (option "fast_path", default on; cells that fit one batch): scoring, resolve and the copy of
// ONE result block (header + grid + scores) are enqueued back to back and the host waits once.  The header says
// whether anything the host would have acted on in between occurred -- a cell the filter could not vouch for, an
// overflow mark, sweeps still changing picks, an unverified speculative pick; if so the call continues on the
// step-by-step path from the first thing that cannot be trusted (lists, or the resolve only), so the result is the
// same either way.  The reference's everyday jobs (~1000 cells, ~200 images) are bound by exactly these round
// trips: the per-cell H2D / memset / sync sequence of CUDAPhotomosaicGenerator.cpp:176-229 is what they replace.
mm_status mm_generate(mm_engine* e, const mm_cell* cells, int n_cells, int grid_w, int grid_h, int diff_type,
                      int repeat_range, int repeat_addition, int64_t* best_fit_out, double* best_score_out,
                      mm_progress_cb progress, void* progress_user, volatile int* cancel)
{
    if (!e) return MM_INVALID_ARGUMENT;
    if (!known_type(diff_type)) { set_err(e, "No function for given type %d", diff_type); return MM_INVALID_ARGUMENT; }
    if (n_cells < 0 || (n_cells > 0 && !cells) || grid_w <= 0 || grid_h <= 0 || !best_fit_out || repeat_range < 0) {
        set_err(e, "mm_generate: bad arguments");
        return MM_INVALID_ARGUMENT;
    }
    mm_status st = check_ready(e);
    if (st) return st;
    if ((st = check_cells(e, cells, n_cells, grid_w, grid_h))) return st;
    HIP_TRY(e, select_device(e), "Failed to select device");
    e->timing = mm_timing{};
    e->syncs = 0;
    e->result_grid_valid = false;

    const int K = std::max(1, mm_candidates_per_cell(e->n_lib, repeat_range, repeat_addition));
    // Short lists of 4 verify every pick of the synthetic benchmark in one round; on content where neighbouring cells want the
    // same few images (photographs) a share of the cells has to be deepened, and whether longer lists or more rounds are cheaper
    // depends on the job: the length follows what the engine's previous generates over the same job MEASURED (learn_list_length
    // below).  As with the filter's item size, only the TIME of a call depends on the history, never its result ("spec_auto" 0 pins it).
    int spec_k = e->spec_k;
    if (e->spec_auto != 0 && spec_k > 0 && e->spec_lib == e->n_lib && e->spec_size == e->lib_size && e->spec_cells == n_cells && e->spec_k_learned > 0)
        spec_k = e->spec_k_learned;
    const int Ks = std::max(1, mm_speculative_candidates(e->n_lib, repeat_range, repeat_addition, spec_k));
    const bool speculate = Ks < K && n_cells > 0;
    // Deepening BY NEED (round 6): on photographs a cell needs 3.8 list entries on average, a tenth of the cells more than 8, none more than 16
    // (tools/list_length_sim.py on config PD) -- and the counting bound's full length is 42 at repeat range 4.  A cell whose short list did
    // not verify is therefore extended to a MIDDLE length first (its filter pass runs against the limit of the Kmid-th seed instead of the
    // 42nd: a fraction of the walks), and to the full length only if that did not verify either.  Rows are K wide and padded with empty
    // entries behind what is filled, so the resolve reads every non-deep cell as a list of <= Kmid entries whatever its level.
    const int Kmid = (speculate && e->deepen_mid > 0 && std::max(e->deepen_mid, 2 * Ks) < K) ? std::max(e->deepen_mid, 2 * Ks) : K;
    const int Kfill = Kmid < K ? Kmid : Ks;        // what the resolve takes as the length of a non-deep list
    // The short lists' length by MEASUREMENT (round 6; until then: "more than 2 % of the cells deepened doubles it").  Longer lists cost every
    // cell's filter limit and fp64 re-score; shorter ones cost deepening rounds, each a resolve and a few host waits.  Which is cheaper
    // depends on the job: photographs at config D's scale run 17.9 ms per generate at 4 entries with deepening by need against 21.3 at
    // 16 entries without a second round; the 704 cells of config P 4.8 against 4.0.  So: a generate that had to deepen more than 2 % of its
    // cells lets the NEXT generate over the same job (library, cell size, cell count) try twice the length, once per length up to four
    // times the base; every generate records its own total time under its length, and once the lengths due for a try have been tried the
    // fastest one is kept.  As everywhere else here, only the TIME of a call depends on the engine's history, never its result.
    auto learn_list_length = [&]() {
        if (!speculate || e->spec_auto == 0 || e->spec_k <= 0) return;
        if (e->spec_lib != e->n_lib || e->spec_size != e->lib_size || e->spec_cells != n_cells) {
            e->spec_lib = e->n_lib; e->spec_size = e->lib_size; e->spec_cells = n_cells; e->spec_k_learned = 0;
            e->spec_ms[0] = e->spec_ms[1] = e->spec_ms[2] = 0.f;
            e->spec_base_runs = 0;
        }
        const int base = e->spec_k;
        const int li = spec_k >= 4 * base ? 2 : (spec_k >= 2 * base ? 1 : 0);
        const float ms = (float)e->timing.total_ms;
        if (ms > 0.f) e->spec_ms[li] = e->spec_ms[li] > 0.f ? std::min(e->spec_ms[li], ms) : ms;
        if (li == 0) e->spec_base_runs += 1;
        const bool many_deepened = (double)e->timing.deepened_cells > 0.02 * n_cells;
        if (many_deepened && li < 2 && (base << (li + 1)) <= 16 && e->spec_ms[li + 1] == 0.f) {
            e->spec_k_learned = base << (li + 1);          // the next length gets its try
        } else if (e->spec_ms[1] > 0.f && e->spec_base_runs < 2) {
            e->spec_k_learned = 0;                          // the base length once more, warm: its first generate was the library's first
        } else {
            int best = -1;
            for (int i = 0; i < 3; ++i)
                if (e->spec_ms[i] > 0.f && (best < 0 || e->spec_ms[i] < e->spec_ms[best])) best = i;
            e->spec_k_learned = best > 0 ? base << best : 0;
        }
    };
    const size_t cells_total = (size_t)grid_w * grid_h;
    HIP_TRY(e, hipEventRecord(e->ev[6], e->stream), "event");
    if ((st = ensure(e, e->recs, (size_t)std::max(n_cells, 1) * sizeof(mm_cell), "Failed to allocate cell records")) ||
        (st = ensure(e, e->cand_i, (size_t)std::max(n_cells, 1) * K * sizeof(int32_t), "Failed to allocate candidates")) ||
        (st = ensure(e, e->cand_s, (size_t)std::max(n_cells, 1) * K * sizeof(double), "Failed to allocate candidates")))
        return st;
    HIP_TRY(e, hipMemcpyAsync(e->recs.p, cells, (size_t)n_cells * sizeof(mm_cell), hipMemcpyHostToDevice, e->stream),
            "Failed to copy cell records to device");
    const bool sequential = repeat_range > 0 && repeat_addition != 0;
    const size_t ring_bytes = sequential ? (size_t)(repeat_range + 1) * grid_w * sizeof(int) : 0;
    const bool want_fast = e->fast_path != 0 && n_cells > 0 && e->use_wave_resolve != 0 && ring_bytes <= 120 * 1024 &&
                           (!sequential || e->use_sweep_resolve != 0);
    // events between the phases: worth their queue slots only where a phase outlasts them by far
    // (measured on config A: ~35 us per call for the nine events, 0.081 -> 0.121 ms per step: on where that is below ~1 %)
    e->phase_events = e->phase_timing > 0 ||
                      (e->phase_timing < 0 && (double)n_cells * e->n_lib * e->lib_size * e->lib_size * (diff_type == MM_CIEDE2000 ? 1.0 : 0.1) >= 2e10);
    e->fast_hdr = nullptr;
    if (want_fast && (st = begin_single_submission(e, n_cells, grid_w, grid_h, speculate))) return st;
    if (n_cells > 0) {
        st = score_cells_device(e, cells, (const mm_cell*)e->recs.p, n_cells, diff_type, Ks, K, (int32_t*)e->cand_i.p,
                                (double*)e->cand_s.p, nullptr, progress, progress_user, cancel, want_fast);
        if (st) { e->fast_hdr = nullptr; return st; }
    }
    if (cancel && *cancel) { e->fast_hdr = nullptr; return MM_CANCELLED; }

    std::vector<uint8_t> deep, insufficient;
    bool have_verdicts = false;            // the result block holds the final state of the lists as they are now
    if (!e->pend.active) e->fast_hdr = nullptr;     // more than one batch: the step-by-step path
    if (e->pend.active) {
        const mm_engine::Pending pd = e->pend;
        e->pend.active = false;
        st = resolve_single_submission(e, (const mm_cell*)e->recs.p, n_cells, grid_w, grid_h, diff_type, repeat_range, repeat_addition,
                                       K, Kfill, speculate, (const int32_t*)e->cand_i.p, (const double*)e->cand_s.p, best_score_out != nullptr,
                                       pd.two_tier ? (const uint8_t*)e->flags.p : nullptr, pd.counted);
        if (st) return st;
        const mm::ResultHeader& h = *(const mm::ResultHeader*)e->pinned;
        const bool untrusted = pd.two_tier && (h.launch_void != 0u || h.edge_pairs > pd.edge_cap);     // voided, or the log overflowed
        account_score_batch(e, pd.two_tier, pd.prune, pd.diff_type, pd.b, h.edge_pairs, h.work, pd.two_tier && h.launch_void != 0u);
        if (progress) progress(n_cells, progress_user);
        const bool lists_ok = h.n_flagged == 0 && !untrusted;
        int settled = -1;                  // first sweep that changed nothing
        for (unsigned int i = 0; i < h.sweeps && settled < 0; ++i)
            if (h.sweep_changes[i] == 0) settled = (int)i;
        int sweeps_before = 0;
        if (lists_ok && h.n_marked == 0 && sequential && settled < 0 &&
            (st = settle_sweeps(e, (const mm_cell*)e->recs.p, n_cells, grid_w, grid_h, diff_type, repeat_range, repeat_addition, K, Kfill, speculate,
                                (const int32_t*)e->cand_i.p, (const double*)e->cand_s.p, best_score_out != nullptr, &settled, &sweeps_before)))
            return st;
        const bool resolved_ok = lists_ok && h.n_marked == 0 && (!sequential || settled >= 0);
        if (!lists_ok) {
            std::vector<uint8_t> flags_host((size_t)n_cells);
            HIP_TRY(e, hipMemcpy(flags_host.data(), e->flags.p, (size_t)n_cells, hipMemcpyDeviceToHost), "flags");
            if ((st = redo_flagged_cells(e, cells, 0, n_cells, flags_host.data(), untrusted, diff_type,
                                         Ks, K, (int32_t*)e->cand_i.p, (double*)e->cand_s.p, nullptr)))
                return st;
        }
        if (resolved_ok) {
            if (e->phase_events) e->timing.resolve_ms += elapsed(e->ev[10], e->ev[11]);
            e->timing.resolve_sweeps += sequential ? sweeps_before + settled + 1 : 0;
            if (sequential && !e->last_front) { if (e->sweeps_lib != e->n_lib || e->sweeps_size != e->lib_size) e->front_learned = false; e->sweeps_learned = sweeps_before + settled + 1; e->sweeps_lib = e->n_lib; e->sweeps_size = e->lib_size; }
            e->timing.exact_rescored += (int64_t)(sequential ? h.sweep_exact[settled] : h.exact_plain);
            e->timing.dense_rows += (int64_t)h.dense_count;
            const char* body = (const char*)e->pinned + sizeof(mm::ResultHeader);
            memcpy(best_fit_out, body, cells_total * sizeof(int64_t));
            if (best_score_out) memcpy(best_score_out, body + cells_total * sizeof(int64_t), cells_total * sizeof(double));
            if (!speculate || h.n_insufficient == 0) {
                e->timing.spec_rounds = 1;
                e->timing.fast_path = 1;
                e->timing.total_ms = elapsed(e->ev[6], e->ev[7]);
                e->timing.pair_evals = count_pair_evals(e, cells, n_cells);
                e->timing.host_syncs = e->syncs;
                learn_list_length();
                return MM_OK;
            }
            have_verdicts = true;          // only the unverified cells are left: straight to the deepening step
        }
    }

    std::vector<mm_cell> sub_cells;
    std::vector<int> sub_rows;
    std::vector<uint8_t> level;            // 0: the cell's list is the short one, 1: it was extended to the middle length
    int extended_since_resolve = 0;        // cells whose lists were extended since the last resolve
    if (speculate) {
        deep.assign((size_t)n_cells, 0);
        level.assign((size_t)n_cells, 0);
        insufficient.assign((size_t)n_cells, 0);
        if ((st = ensure(e, e->deep, (size_t)n_cells, "Failed to allocate list flags")) ||
            (st = ensure(e, e->insufficient, (size_t)n_cells, "Failed to allocate list flags")))
            return st;
    }
    for (int round = 0;; ++round) {
        if (!(round == 0 && have_verdicts)) {
            // The resolve of a later round as ONE submission too (result block initialised, eight sweeps, one copy, one wait)
            // where the first round took that path: the step-by-step resolve asks the host about ten times per round -- on
            // photographic content, where neighbours collide and the lists are deepened over several rounds (bench.py
            // --config P: 4 rounds, 42 host waits, 2.3 of 7.3 ms), that was the largest phase of the step.  Anything the
            // block cannot vouch for (an overflow mark, sweeps that did not settle) falls through to the step-by-step form.
            bool resolved = false;
            if (want_fast) {
                // (warm: the previous round of THIS generate left its fix point in the block, and only a few cells were extended since -- what
                // they displace stays local and the sweeps settle in a handful of passes: config P's round after one extended cell 0.56 -> 0.32 ms;
                // after the 1176 cells of config PD's first round eight sweeps do not settle and only delay the front form: 2.67 -> 2.94 ms)
                const bool warm = sequential && e->result_grid_valid && e->warm_rounds != 0 && extended_since_resolve <= std::max(8, n_cells / 256);
                if ((st = begin_single_submission(e, n_cells, grid_w, grid_h, speculate, warm))) return st;
                if (speculate)      // (after the block's initialisation, which clears the flag arrays, in stream order)
                    HIP_TRY(e, hipMemcpyAsync(e->deep.p, deep.data(), (size_t)n_cells, hipMemcpyHostToDevice, e->stream), "Failed to copy list flags");
                st = resolve_single_submission(e, (const mm_cell*)e->recs.p, n_cells, grid_w, grid_h, diff_type, repeat_range, repeat_addition,
                                               K, Kfill, speculate, (const int32_t*)e->cand_i.p, (const double*)e->cand_s.p, best_score_out != nullptr,
                                               nullptr, false, false, 0, warm);
                if (st) return st;
                const mm::ResultHeader& h = *(const mm::ResultHeader*)e->pinned;
                int settled = -1, sweeps_before = 0;
                for (unsigned int i = 0; i < h.sweeps && settled < 0; ++i)
                    if (h.sweep_changes[i] == 0) settled = (int)i;
                if (h.n_marked == 0 && sequential && settled < 0 &&
                    (st = settle_sweeps(e, (const mm_cell*)e->recs.p, n_cells, grid_w, grid_h, diff_type, repeat_range, repeat_addition, K, Kfill, speculate,
                                        (const int32_t*)e->cand_i.p, (const double*)e->cand_s.p, best_score_out != nullptr, &settled, &sweeps_before)))
                    return st;
                if (h.n_marked == 0 && (!sequential || settled >= 0)) {
                    if (e->phase_events) e->timing.resolve_ms += elapsed(e->ev[10], e->ev[11]);
                    e->timing.resolve_sweeps += sequential ? sweeps_before + settled + 1 : 0;
                    if (sequential && !e->last_front) { if (e->sweeps_lib != e->n_lib || e->sweeps_size != e->lib_size) e->front_learned = false; e->sweeps_learned = sweeps_before + settled + 1; e->sweeps_lib = e->n_lib; e->sweeps_size = e->lib_size; }
                    e->timing.exact_rescored += (int64_t)(sequential ? h.sweep_exact[settled] : h.exact_plain);
                    e->timing.dense_rows += (int64_t)h.dense_count;
                    const char* body = (const char*)e->pinned + sizeof(mm::ResultHeader);
                    memcpy(best_fit_out, body, cells_total * sizeof(int64_t));
                    if (best_score_out) memcpy(best_score_out, body + cells_total * sizeof(int64_t), cells_total * sizeof(double));
                    resolved = true;
                }
            }
            if (!resolved) {
                if (speculate) {
                    HIP_TRY(e, hipMemcpyAsync(e->deep.p, deep.data(), (size_t)n_cells, hipMemcpyHostToDevice, e->stream), "Failed to copy list flags");
                    HIP_TRY(e, hipMemsetAsync(e->insufficient.p, 0, (size_t)n_cells, e->stream), "Failed to clear list flags");
                }
                st = resolve_device(e, (const mm_cell*)e->recs.p, n_cells, grid_w, grid_h, diff_type, repeat_range,
                                    repeat_addition, K, (const int32_t*)e->cand_i.p, (const double*)e->cand_s.p,
                                    best_fit_out, best_score_out,
                                    speculate ? (const uint8_t*)e->deep.p : nullptr, speculate ? (uint8_t*)e->insufficient.p : nullptr, Kfill);
                if (st) return st;
            }
        }
        e->timing.spec_rounds = round + 1;
        extended_since_resolve = 0;
        if (!speculate) break;
        HIP_TRY(e, hipMemcpy(insufficient.data(), e->insufficient.p, (size_t)n_cells, hipMemcpyDeviceToHost), "Error copying list flags to host");
        // unverified cells: those still on their short list go to the middle length, those already there to the full one
        bool any = false;
        const std::vector<uint8_t> level_before = level;    // (a cell extended to the middle length in this round is not also sent on to the full one)
        for (int target = 0; target < 2; ++target) {       // 0: short -> middle (only if there is a middle length), 1: -> full
            if (target == 0 && Kmid >= K) continue;
            sub_cells.clear();
            sub_rows.clear();
            for (int c = 0; c < n_cells; ++c) {
                if (!insufficient[(size_t)c] || deep[(size_t)c]) continue;
                const bool at_mid = level_before[(size_t)c] != 0;
                if (target == 0 ? !at_mid : (at_mid || Kmid >= K)) {
                    sub_cells.push_back(cells[c]);
                    sub_rows.push_back(c);
                    if (target == 0) level[(size_t)c] = 1; else deep[(size_t)c] = 1;
                }
            }
            if (sub_cells.empty()) continue;
            any = true;
            if (cancel && *cancel) return MM_CANCELLED;
            const int ns = (int)sub_cells.size();
            if ((st = ensure(e, e->spec_recs, (size_t)ns * sizeof(mm_cell), "Failed to allocate cell records"))) return st;
            HIP_TRY(e, hipMemcpyAsync(e->spec_recs.p, sub_cells.data(), (size_t)ns * sizeof(mm_cell), hipMemcpyHostToDevice, e->stream),
                    "Failed to copy cell records to device");
            // (a handful of cells is latency either way: sending them down the fp64-only path instead -- three launches for six -- was
            // measured no faster, config P 5.5 -> 5.6 ms: one cell's 1024 fp64 rows are 64 workgroups of a kernel laid out for thousands)
            st = score_cells_device(e, sub_cells.data(), (const mm_cell*)e->spec_recs.p, ns, diff_type, target == 0 ? Kmid : K, K,
                                    (int32_t*)e->cand_i.p, (double*)e->cand_s.p, sub_rows.data(), nullptr, nullptr, cancel);
            if (st) return st;
            e->timing.deepened_cells += ns;
            extended_since_resolve += ns;
        }
        if (!any) break;
    }
    HIP_TRY(e, hipEventRecord(e->ev[7], e->stream), "event");
    HIP_TRY(e, sync_stream(e), "Failed to synchronise stream");
    e->timing.total_ms = elapsed(e->ev[6], e->ev[7]);
    e->timing.pair_evals = count_pair_evals(e, cells, n_cells);
    e->timing.host_syncs = e->syncs;
    learn_list_length();
    return MM_OK;
}

mm_status mm_set_library_bgr(mm_engine* e, const uint8_t* images, int n_images, int size)
{
    if (!e) return MM_INVALID_ARGUMENT;
    if (!images || n_images <= 0 || size <= 0 || size > 4096) {
        set_err(e, "mm_set_library_bgr: need n_images > 0 and 0 < size <= 4096 (got %d, %d)", n_images, size);
        return MM_INVALID_ARGUMENT;
    }
    HIP_TRY(e, select_device(e), "Failed to select device");
    const size_t bytes = (size_t)n_images * size * size * 3;
    mm_status st = ensure(e, e->lib_bgr, bytes, "Failed to allocate library images");
    if (st) return st;
    HIP_TRY(e, hipMemcpy(e->lib_bgr.p, images, bytes, hipMemcpyHostToDevice), "Failed to copy library image to device");
    e->n_lib_bgr = n_images;
    e->lib_bgr_size = size;
    return MM_OK;
}

mm_status mm_shrink_library_bgr(mm_engine* e, int factor)
{
    if (!e) return MM_INVALID_ARGUMENT;
    if (!e->lib_bgr.p || e->n_lib_bgr <= 0) { set_err(e, "mm_shrink_library_bgr: no u8 library staged"); return MM_NOT_READY; }
    if (factor < 1 || e->lib_bgr_size % factor != 0) {
        set_err(e, "mm_shrink_library_bgr: factor %d does not divide the image size %d", factor, e->lib_bgr_size);
        return MM_INVALID_ARGUMENT;
    }
    if (factor == 1) return MM_OK;
    HIP_TRY(e, select_device(e), "Failed to select device");
    const int so = e->lib_bgr_size / factor;
    const size_t out_px = (size_t)e->n_lib_bgr * so * so;
    mm_status st = ensure(e, e->stage_u8, out_px * 3, "Failed to allocate library images");
    if (st) return st;
    const unsigned blocks = (unsigned)std::min<size_t>((out_px + mm::kBlock - 1) / mm::kBlock, 65536);
    mm::shrink_u8_kernel<<<std::max(blocks, 1u), mm::kBlock, 0, e->stream>>>(
        (const uint8_t*)e->lib_bgr.p, (uint8_t*)e->stage_u8.p, e->n_lib_bgr, e->lib_bgr_size, factor);
    HIP_TRY(e, hipGetLastError(), "Error during library resize kernel");
    HIP_TRY(e, sync_stream(e), "Failed to synchronise stream");
    std::swap(e->lib_bgr, e->stage_u8);
    e->lib_bgr_size = so;
    return MM_OK;
}

mm_status mm_build_photomosaic(mm_engine* e, const mm_cell* cells, int n_cells, int grid_w, int grid_h,
                               const int64_t* best_fit, const uint8_t background[4], int rows, int cols,
                               uint8_t* mosaic_out)
{
    if (!mosaic_out) { set_err(e, "mm_build_photomosaic: bad arguments"); return e ? MM_INVALID_ARGUMENT : MM_INVALID_ARGUMENT; }
    return mm_build_photomosaic_step(e, cells, n_cells, grid_w, grid_h, best_fit, background, rows, cols, 0, mosaic_out);
}

mm_status mm_build_photomosaic_step(mm_engine* e, const mm_cell* cells, int n_cells, int grid_w, int grid_h,
                                    const int64_t* best_fit, const uint8_t background[4], int rows, int cols,
                                    int step_index, uint8_t* mosaic_out)
{
    if (!e) return MM_INVALID_ARGUMENT;
    if (n_cells < 0 || (n_cells > 0 && !cells) || grid_w <= 0 || grid_h <= 0 || !best_fit || !background ||
        rows <= 0 || cols <= 0 || step_index < 0) {
        set_err(e, "mm_build_photomosaic: bad arguments");
        return MM_INVALID_ARGUMENT;
    }
    if (step_index > 0 && (!e->covered.p || e->mosaic_rows != rows || e->mosaic_cols != cols)) {
        set_err(e, "mm_build_photomosaic_step: step %d without a step 0 of the same %dx%d mosaic", step_index, cols, rows);
        return MM_NOT_READY;
    }
    if (!e->lib_bgr.p || e->n_lib_bgr <= 0 || !e->masks.p || e->mask_size <= 0) {
        set_err(e, "u8 library (mm_set_library_bgr) and masks must be staged before building the photomosaic");
        return MM_NOT_READY;
    }
    if (e->lib_bgr_size != e->mask_size) {
        set_err(e, "library image size %d != mask size %d", e->lib_bgr_size, e->mask_size);
        return MM_INVALID_ARGUMENT;
    }
    const int size = e->lib_bgr_size, P = size * size, ppad = round_up(P, 256);
    for (int i = 0; i < n_cells; ++i) {
        const mm_cell& c = cells[i];
        const bool ok = c.row_min >= 0 && c.row_min < c.row_max && c.row_max <= size && c.col_min >= 0 &&
                        c.col_min < c.col_max && c.col_max <= size && c.mask_id >= 0 && c.mask_id < 4 &&
                        (long long)c.y + c.row_min >= 0 && (long long)c.y + c.row_max <= rows &&
                        (long long)c.x + c.col_min >= 0 && (long long)c.x + c.col_max <= cols &&
                        c.grid_x >= 0 && c.grid_x < grid_w && c.grid_y >= 0 && c.grid_y < grid_h;
        if (!ok) { set_err(e, "mm_build_photomosaic: cell %d does not fit the %dx%d image / %dx%d grid", i, cols, rows, grid_w, grid_h); return MM_INVALID_ARGUMENT; }
    }
    HIP_TRY(e, select_device(e), "Failed to select device");
    const size_t n_pixels = (size_t)rows * cols, grid_cells = (size_t)grid_w * grid_h;
    mm_status st;
    if ((st = ensure(e, e->recs, (size_t)std::max(n_cells, 1) * sizeof(mm_cell), "Failed to allocate cell records")) ||
        (st = ensure(e, e->grid, grid_cells * sizeof(long long), "Failed to allocate best fits")) ||
        (st = ensure(e, e->owner, n_pixels * sizeof(unsigned int), "Failed to allocate owner map")) ||
        (st = ensure(e, e->mosaic, n_pixels * sizeof(uint32_t), "Failed to allocate photomosaic")) ||
        (st = ensure(e, e->covered, n_pixels, "Failed to allocate photomosaic mask")))
        return st;
    e->mosaic_rows = rows;
    e->mosaic_cols = cols;
    HIP_TRY(e, hipEventRecord(e->ev[6], e->stream), "event");
    HIP_TRY(e, hipMemcpyAsync(e->recs.p, cells, (size_t)n_cells * sizeof(mm_cell), hipMemcpyHostToDevice, e->stream),
            "Failed to copy cell records to device");
    HIP_TRY(e, hipMemcpyAsync(e->grid.p, best_fit, grid_cells * sizeof(long long), hipMemcpyHostToDevice, e->stream),
            "Failed to copy best fits to device");
    HIP_TRY(e, hipMemsetAsync(e->owner.p, 0, n_pixels * sizeof(unsigned int), e->stream), "Failed to clear owner map");
    HIP_TRY(e, hipEventRecord(e->ev[4], e->stream), "event");
    if (n_cells > 0) {
        mm::mosaic_owner_kernel<<<dim3((P + mm::kBlock - 1) / mm::kBlock, n_cells), mm::kBlock, 0, e->stream>>>(
            (const mm_cell*)e->recs.p, (const long long*)e->grid.p, grid_w, e->n_lib_bgr, (const uint8_t*)e->masks.p,
            size, P, ppad, cols, (unsigned int*)e->owner.p);
        HIP_TRY(e, hipGetLastError(), "Error during photomosaic owner kernel");
    }
    const uint32_t bg = (uint32_t)background[0] | ((uint32_t)background[1] << 8) | ((uint32_t)background[2] << 16) |
                        ((uint32_t)background[3] << 24);
    const unsigned blocks = (unsigned)std::min<size_t>((n_pixels + mm::kBlock - 1) / mm::kBlock, 16384);
    mm::mosaic_compose_kernel<<<blocks, mm::kBlock, 0, e->stream>>>(
        (const unsigned int*)e->owner.p, (const mm_cell*)e->recs.p, (const long long*)e->grid.p, grid_w,
        (const uint8_t*)e->lib_bgr.p, size, cols, n_pixels, bg, (uint32_t*)e->mosaic.p,
        (uint8_t*)e->covered.p, step_index == 0);
    HIP_TRY(e, hipGetLastError(), "Error during photomosaic compose kernel");
    HIP_TRY(e, hipEventRecord(e->ev[5], e->stream), "event");
    if (mosaic_out)
        HIP_TRY(e, hipMemcpyAsync(mosaic_out, e->mosaic.p, n_pixels * 4, hipMemcpyDeviceToHost, e->stream),
                "Error copying photomosaic to host");
    HIP_TRY(e, hipEventRecord(e->ev[7], e->stream), "event");
    HIP_TRY(e, sync_stream(e), "Failed to synchronise stream");
    e->timing = mm_timing{};
    e->timing.build_ms = elapsed(e->ev[4], e->ev[5]);
    e->timing.total_ms = elapsed(e->ev[6], e->ev[7]);
    return MM_OK;
}

mm_status mm_set_main_image_bgr(mm_engine* e, const uint8_t* bgr, int rows, int cols)
{
    if (!e) return MM_INVALID_ARGUMENT;
    if (!bgr || rows <= 0 || cols <= 0) {
        set_err(e, "mm_set_main_image_bgr: need rows, cols > 0 (got %d, %d)", rows, cols);
        return MM_INVALID_ARGUMENT;
    }
    HIP_TRY(e, select_device(e), "Failed to select device");
    const size_t bytes = (size_t)rows * cols * 3;
    mm_status st = ensure(e, e->main_bgr, bytes, "Failed to allocate main image");
    if (st) return st;
    HIP_TRY(e, hipMemcpy(e->main_bgr.p, bgr, bytes, hipMemcpyHostToDevice), "Failed to copy main image to device");
    e->main_rows = rows;
    e->main_cols = cols;
    return MM_OK;
}

// GridGenerator::findCellState's entropy test (Grid/GridGenerator.cpp:143-189) for a batch of cells of one size step.
mm_status mm_cell_entropy(mm_engine* e, const mm_cell* cells, int n_cells, int normal_size, uint8_t* exceeded_out, double* entropy_out)
{
    if (!e) return MM_INVALID_ARGUMENT;
    if (n_cells < 0 || (n_cells > 0 && (!cells || !exceeded_out)) || normal_size <= 0) { set_err(e, "mm_cell_entropy: bad arguments"); return MM_INVALID_ARGUMENT; }
    if (!e->main_bgr.p || e->main_rows <= 0 || !e->masks.p || e->mask_size <= 0) { set_err(e, "the u8 main image (mm_set_main_image_bgr) and the step's detail masks must be staged first"); return MM_NOT_READY; }
    const int ds = e->mask_size;
    if (ds > normal_size) { set_err(e, "mm_cell_entropy: detail masks of %d px against a normal cell of %d px", ds, normal_size); return MM_INVALID_ARGUMENT; }
    for (int i = 0; i < n_cells; ++i) {
        const mm_cell& c = cells[i];
        const bool ok = c.row_min >= 0 && c.row_min < c.row_max && c.row_max <= ds && c.col_min >= 0 && c.col_min < c.col_max &&
                        c.col_max <= ds && c.mask_id >= 0 && c.mask_id < 4 &&
                        std::abs((long long)c.x) < (1LL << 28) && std::abs((long long)c.y) < (1LL << 28);
        if (!ok) { set_err(e, "mm_cell_entropy: cell %d does not fit the %d px detail mask", i, ds); return MM_INVALID_ARGUMENT; }
        // a shrink only: the visible part is never smaller than the bounded mask for detail <= 100 %
        const long long vr = std::min<long long>(std::max<long long>((long long)c.y + normal_size, 0), e->main_rows) - std::min<long long>(std::max<long long>(c.y, 0), e->main_rows);
        const long long vc = std::min<long long>(std::max<long long>((long long)c.x + normal_size, 0), e->main_cols) - std::min<long long>(std::max<long long>(c.x, 0), e->main_cols);
        if (vr > 0 && vc > 0 && (vr < c.row_max - c.row_min || vc < c.col_max - c.col_min)) {
            set_err(e, "mm_cell_entropy: cell %d would be ENLARGED (%lldx%lld visible pixels to a %dx%d mask): INTER_CUBIC is outside the hot path",
                    i, vc, vr, c.col_max - c.col_min, c.row_max - c.row_min);
            return MM_INVALID_ARGUMENT;
        }
    }
    if (n_cells == 0) return MM_OK;
    HIP_TRY(e, select_device(e), "Failed to select device");
    mm_status st;
    if ((st = ensure(e, e->recs, (size_t)n_cells * sizeof(mm_cell), "Failed to allocate cell records")) ||
        (st = ensure(e, e->exceeded_out, (size_t)n_cells, "Failed to allocate entropy verdicts")) ||
        (st = ensure(e, e->entropy_out, (size_t)n_cells * sizeof(double), "Failed to allocate entropies")))
        return st;
    HIP_TRY(e, hipMemcpyAsync(e->recs.p, cells, (size_t)n_cells * sizeof(mm_cell), hipMemcpyHostToDevice, e->stream),
            "Failed to copy cell records to device");
    const int ppad = round_up(ds * ds, 256);
    mm::cell_entropy_kernel<<<n_cells, mm::kBlock, 0, e->stream>>>(
        (const mm_cell*)e->recs.p, n_cells, (const uint8_t*)e->main_bgr.p, e->main_rows, e->main_cols, normal_size,
        (const uint8_t*)e->masks.p, ds, ppad, (uint8_t*)e->exceeded_out.p, (double*)e->entropy_out.p);
    HIP_TRY(e, hipGetLastError(), "Error during cell entropy kernel");
    HIP_TRY(e, hipMemcpyAsync(exceeded_out, e->exceeded_out.p, (size_t)n_cells, hipMemcpyDeviceToHost, e->stream), "Error copying entropy verdicts to host");
    if (entropy_out)
        HIP_TRY(e, hipMemcpyAsync(entropy_out, e->entropy_out.p, (size_t)n_cells * sizeof(double), hipMemcpyDeviceToHost, e->stream),
                "Error copying entropies to host");
    HIP_TRY(e, sync_stream(e), "Failed to synchronise stream");
    return MM_OK;
}

// Diagnostics: pixel groups visited by the filter kernel, groups a full visit would take, candidates
// dropped by the bound -- of the last filter launch (always counted; mm_timing sums them over a generate).
mm_status mm_debug_visit_counters(mm_engine* e, unsigned long long out[3])
{
    if (!e || !out || !e->edge_count.p) return MM_INVALID_ARGUMENT;
    unsigned int raw[8] = {0};
    HIP_TRY(e, hipMemcpy(raw, e->edge_count.p, sizeof raw, hipMemcpyDeviceToHost), "counters");
    memcpy(out, raw + 2, 3 * sizeof(unsigned long long));
#ifdef MM_FILTER_STAMPS
    {   // instrumented build: the cycle accounts of filter_kernel (words 24..31, units of 1024 cycles)
        unsigned int st[8] = {0};
        HIP_TRY(e, hipMemcpy(st, (unsigned int*)e->edge_count.p + 24, sizeof st, hipMemcpyDeviceToHost), "stamps");
        fprintf(stderr, "stamps wait %u fetch %u stats %u stage0 %u stage1 %u stage2 %u longest lifetime %u sum of lifetimes / 16 %u\n", st[0], st[1], st[2], st[3], st[4], st[5], st[6], st[7]);
    }
#endif
    return MM_OK;
}

// Calibration: GB/s of a coalesced float4 read stream over `bytes` of device memory (larger than
// the 256 MiB Infinity Cache to see HBM), best of `repeats` launches timed with HIP events.
mm_status mm_debug_stream_read(mm_engine* e, size_t bytes, int repeats, double* gb_per_s)
{
    if (!e || !gb_per_s || bytes < (1u << 20) || repeats < 1) return MM_INVALID_ARGUMENT;
    HIP_TRY(e, select_device(e), "Failed to select device");
    const size_t n_vec = bytes / 16;
    const int blocks = 256 * 16;                                  // 16 workgroups of 4 waves per CU
    DevBuf src, sink;
    mm_status st;
    if ((st = ensure(e, src, n_vec * 16, "Failed to allocate stream buffer")) ||
        (st = ensure(e, sink, (size_t)blocks * mm::kBlock * sizeof(float), "Failed to allocate stream sink"))) {
        release(src); release(sink);
        return st;
    }
    hipError_t code = hipMemsetAsync(src.p, 0, n_vec * 16, e->stream);
    double best_ms = 1e30;
    for (int r = 0; r <= repeats && code == hipSuccess; ++r) {     // launch 0 warms up
        if ((code = hipEventRecord(e->ev[4], e->stream)) != hipSuccess) break;
        mm::stream_read_kernel<<<blocks, mm::kBlock, 0, e->stream>>>((const float4*)src.p, n_vec, (float*)sink.p);
        if ((code = hipGetLastError()) != hipSuccess) break;
        if ((code = hipEventRecord(e->ev[5], e->stream)) != hipSuccess) break;
        if ((code = sync_stream(e)) != hipSuccess) break;
        if (r > 0) best_ms = std::min(best_ms, (double)elapsed(e->ev[4], e->ev[5]));
    }
    release(src); release(sink);
    if (code != hipSuccess) return hip_fail(e, code, "Error during stream read calibration", __FILE__, __LINE__);
    *gb_per_s = (double)(n_vec * 16) / (best_ms * 1e-3) / 1e9;
    return MM_OK;
}

// Backs filter_band(): a structured sweep (hue x hue x chroma x chroma x lightness, and both sides of
// the opposite-hue discontinuity) of the fp32 CIEDE2000 filter form against the fp64 form, generated
// and reduced on the device.  out[0..3] = max |err|, max (|err| - rel v), max |err| / (px_abs + rel v)
// -- the share of the trusted band the worst pair uses -- and the largest fp64 value met;
// out[4], out[5] = pairs handed to the fp64 form / pairs measured.
mm_status mm_debug_filter_band_sweep(mm_engine* e, int mode, uint64_t seed, uint64_t n_pairs, double out[6])
{
    if (!e || !out || mode < 0 || mode > 3 || n_pairs == 0) return MM_INVALID_ARGUMENT;       // (modes 2, 3: the packed-fp16 pre-tier's form against its own band)
    HIP_TRY(e, select_device(e), "Failed to select device");
    mm_status st = ensure_hue_lut(e);
    if (st) return st;
    DevBuf buf;
    if ((st = ensure(e, buf, 6 * sizeof(unsigned long long), "Failed to allocate sweep result"))) return st;
    const mm::FilterBand band = filter_band(MM_CIEDE2000, 0);
    // the per-pair part of the band: px_abs absolute, 6e-5 relative (the rest of band.rel is the accumulation's)
    // (modes 2 / 3 measure against the pre-tier's own band; "dbg_pre_abs_1e4" / "dbg_pre_rel_1e6" try another one: how the band was sized)
    const bool own = (mode & 2) != 0 && (e->dbg_pre_abs > 0.0f || e->dbg_pre_rel > 0.0f);
    hipError_t code = mm::launch_band_sweep(own ? mode | 4 : mode, seed, n_pairs, own ? e->dbg_pre_abs : band.px_abs, own ? e->dbg_pre_rel : 6e-5f,
                                            (const float2*)e->hue_lut.p, (unsigned long long*)buf.p, e->stream);
    unsigned long long raw[6] = {0};
    if (code == hipSuccess) code = hipMemcpyAsync(raw, buf.p, sizeof raw, hipMemcpyDeviceToHost, e->stream);
    if (code == hipSuccess) code = sync_stream(e);
    release(buf);
    if (code != hipSuccess) return hip_fail(e, code, "Error during filter band sweep", __FILE__, __LINE__);
    for (int k = 0; k < 4; ++k) memcpy(&out[k], &raw[k], sizeof(double));
    out[4] = (double)raw[4];
    out[5] = (double)raw[5];
    return MM_OK;
}

mm_status mm_set_option(mm_engine* e, const char* name, int value)
{
    if (!e || !name) return MM_INVALID_ARGUMENT;
    if (strcmp(name, "filter") == 0) { e->use_filter = value != 0; return MM_OK; }
    if (strcmp(name, "prune") == 0) { e->use_prune = value != 0; return MM_OK; }
    if (strcmp(name, "wave_resolve") == 0) { e->use_wave_resolve = value != 0; return MM_OK; }
    if (strcmp(name, "row_pipeline") == 0) { e->use_row_pipeline = value != 0; return MM_OK; }
    if (strcmp(name, "front_resolve") == 0) { e->use_front_resolve = value != 0; e->front_learned = false; return MM_OK; }
    if (strcmp(name, "sweep_resolve") == 0) { e->use_sweep_resolve = value != 0; return MM_OK; }
    if (strcmp(name, "max_sweeps") == 0) { if (value < 1 || value > 4096) return MM_INVALID_ARGUMENT; e->max_sweeps = value; return MM_OK; }
    if (strcmp(name, "lds_filter") == 0) { e->use_lds_filter = value < 0 ? -1 : (value != 0); return MM_OK; }
    if (strcmp(name, "spec_k") == 0) { e->spec_k = std::max(0, value); e->spec_k_learned = 0; e->spec_cells = 0; return MM_OK; }
    if (strcmp(name, "spec_auto") == 0) { e->spec_auto = value != 0; e->spec_k_learned = 0; e->spec_cells = 0; return MM_OK; }
    if (strcmp(name, "deepen_mid") == 0) { e->deepen_mid = std::max(0, value); return MM_OK; }
    if (strcmp(name, "warm_rounds") == 0) { e->warm_rounds = value != 0; return MM_OK; }
    if (strcmp(name, "seeds") == 0) { e->extra_seeds = std::max(-1, value); return MM_OK; }
    if (strcmp(name, "lds_pad_kb") == 0) { e->lds_pad_kb = std::min(120, std::max(0, value)); return MM_OK; }
    if (strcmp(name, "tile") == 0) { e->tile_candidates = std::max(0, value); return MM_OK; }
    if (strcmp(name, "max_batch") == 0) { e->max_batch_cells = std::max(0, value); return MM_OK; }
    if (strcmp(name, "two_level") == 0) { e->two_level = value != 0; return MM_OK; }
    if (strcmp(name, "sort_groups") == 0) { e->sort_groups = std::min(64, std::max(-1, value)); return MM_OK; }
    if (strcmp(name, "interleave") == 0) { e->use_interleave = value != 0; return MM_OK; }
    if (strcmp(name, "xcd_aware") == 0) { e->xcd_aware = value != 0; return MM_OK; }
    if (strcmp(name, "cell_order") == 0) { e->long_items_first = value != 0; return MM_OK; }
    if (strcmp(name, "probe_prepared") == 0) { e->probe_prepared = value == 2 || value == 3 ? value : value != 0; return MM_OK; }
    if (strcmp(name, "check_every") == 0) { e->check_every = std::min(64, std::max(0, value)); return MM_OK; }
    if (strcmp(name, "item_chunk") == 0) { e->item_chunk = std::max(0, value); return MM_OK; }
    if (strcmp(name, "xcd_cells") == 0) { e->xcd_cells = std::max(0, value); return MM_OK; }
    if (strcmp(name, "band_guard") == 0) { e->band_guard_every = std::max(0, value); return MM_OK; }
    if (strcmp(name, "euclid_stage1") == 0) { e->euclid_stage1 = value != 0; return MM_OK; }
    if (strcmp(name, "stage0") == 0) { e->stage0 = value != 0; return MM_OK; }
    if (strcmp(name, "dbg_pre_abs_1e4") == 0) { e->dbg_pre_abs = 1e-4f * (float)std::max(0, value); return MM_OK; }
    if (strcmp(name, "dbg_pre_rel_1e6") == 0) { e->dbg_pre_rel = 1e-6f * (float)std::max(0, value); return MM_OK; }
    if (strcmp(name, "spin_wait_us") == 0) { e->spin_wait_us = std::max(0, std::min(value, 1000000)); return MM_OK; }
    if (strcmp(name, "audit") == 0) {
        int n = std::max(0, std::min(value, 1 << 20));
        while (n & (n - 1)) n &= n - 1;             // the largest power of two not above it
        e->audit = n;
        return MM_OK;
    }
    if (strcmp(name, "dbg_pre_band_pct") == 0) { e->pre_band_pct = value; return MM_OK; }
    if (strcmp(name, "pretier_band") == 0) { e->pretier_band = value != 0; e->pre_auto = mm_engine::PreAuto{}; return MM_OK; }
    if (strcmp(name, "dbg_bound_pct") == 0) { e->bound_pct = std::max(0, value); return MM_OK; }
    if (strcmp(name, "pretier") == 0) { e->pretier = value < 0 ? -1 : (value != 0); e->pre_auto = mm_engine::PreAuto{}; return MM_OK; }
    if (strcmp(name, "tiny") == 0) { e->tiny_path = value != 0; return MM_OK; }
    if (strcmp(name, "item_auto") == 0) { e->item_auto = value != 0; return MM_OK; }
    if (strcmp(name, "fused") == 0) { e->fused = value != 0; return MM_OK; }
    if (strcmp(name, "fast_path") == 0) { e->fast_path = value != 0; return MM_OK; }
    if (strcmp(name, "phase_timing") == 0) { e->phase_timing = value < 0 ? -1 : (value != 0); return MM_OK; }
    if (strcmp(name, "fused_seeds") == 0) { e->fused_seed_cells = std::max(0, value); return MM_OK; }
    if (strcmp(name, "prune_min_kpairs") == 0) { e->prune_min_pairs = value < 0 ? -1 : value; return MM_OK; }
    if (strcmp(name, "band_pct") == 0) { e->band_pct = std::min(1000, std::max(-1000, value)); return MM_OK; }
    set_err(e, "mm_set_option: unknown option '%s'", name);
    return MM_INVALID_ARGUMENT;
}

mm_status mm_get_timing(const mm_engine* e, mm_timing* out)
{
    if (!e || !out) return MM_INVALID_ARGUMENT;
    *out = e->timing;
    return MM_OK;
}

mm_status mm_colour_diff_batch(mm_engine* e, int diff_type, const float* first, const float* second,
                               size_t n, double* out)
{
    if (!e) return MM_INVALID_ARGUMENT;
    if (!known_type(diff_type)) { set_err(e, "No function for given type %d", diff_type); return MM_INVALID_ARGUMENT; }
    if (n == 0) return MM_OK;
    if (!first || !second || !out) { set_err(e, "mm_colour_diff_batch: null pointer"); return MM_INVALID_ARGUMENT; }
    HIP_TRY(e, select_device(e), "Failed to select device");
    DevBuf a, b, o;
    mm_status st;
    if ((st = ensure(e, a, n * 3 * sizeof(float), "Failed to allocate pairs")) ||
        (st = ensure(e, b, n * 3 * sizeof(float), "Failed to allocate pairs")) ||
        (st = ensure(e, o, n * sizeof(double), "Failed to allocate differences"))) {
        release(a); release(b); release(o);
        return st;
    }
    hipError_t code = hipMemcpyAsync(a.p, first, n * 3 * sizeof(float), hipMemcpyHostToDevice, e->stream);
    if (code == hipSuccess) code = hipMemcpyAsync(b.p, second, n * 3 * sizeof(float), hipMemcpyHostToDevice, e->stream);
    if (code == hipSuccess) {
        const int blocks = (int)std::min<size_t>((n + mm::kBlock - 1) / mm::kBlock, 8192);
        by_type(diff_type, [&](auto T) {
            if (e->probe_prepared == 3)        // diagnostics: the parallel sums' algebraic form, to be measured against the reference form
                mm::pair_diff_kernel<decltype(T)::value, true><<<blocks, mm::kBlock, 0, e->stream>>>(
                    (const float*)a.p, (const float*)b.p, n, (double*)o.p);
            else
                mm::pair_diff_kernel<decltype(T)::value, false><<<blocks, mm::kBlock, 0, e->stream>>>(
                    (const float*)a.p, (const float*)b.p, n, (double*)o.p);
            return MM_OK;
        });
        code = hipGetLastError();
    }
    if (code == hipSuccess) code = hipMemcpyAsync(out, o.p, n * sizeof(double), hipMemcpyDeviceToHost, e->stream);
    if (code == hipSuccess) code = sync_stream(e);
    release(a); release(b); release(o);
    if (code != hipSuccess) return hip_fail(e, code, "Error during colour difference kernel", __FILE__, __LINE__);
    return MM_OK;
}

mm_status mm_filter_diff_batch(mm_engine* e, int diff_type, const float* first, const float* second,
                               size_t n, float* out)
{
    if (!e) return MM_INVALID_ARGUMENT;
    if (!known_type(diff_type)) { set_err(e, "No function for given type %d", diff_type); return MM_INVALID_ARGUMENT; }
    if (n == 0) return MM_OK;
    if (!first || !second || !out) { set_err(e, "mm_filter_diff_batch: null pointer"); return MM_INVALID_ARGUMENT; }
    HIP_TRY(e, select_device(e), "Failed to select device");
    DevBuf a, b, o;
    mm_status st;
    if ((st = ensure(e, a, n * 3 * sizeof(float), "Failed to allocate pairs")) ||
        (st = ensure(e, b, n * 3 * sizeof(float), "Failed to allocate pairs")) ||
        (st = ensure(e, o, n * sizeof(float), "Failed to allocate differences"))) {
        release(a); release(b); release(o);
        return st;
    }
    if ((st = ensure_hue_lut(e))) { release(a); release(b); release(o); return st; }
    hipError_t code = hipMemcpyAsync(a.p, first, n * 3 * sizeof(float), hipMemcpyHostToDevice, e->stream);
    if (code == hipSuccess) code = hipMemcpyAsync(b.p, second, n * 3 * sizeof(float), hipMemcpyHostToDevice, e->stream);
    if (code == hipSuccess)
        code = mm::launch_pair_fast(diff_type == MM_CIEDE2000 ? mm::CIEDE2000 : mm::RGB_EUCLIDEAN,
                                    (const float*)a.p, (const float*)b.p, n, (float*)o.p, 1 | (e->probe_prepared == 2 ? 4 : e->probe_prepared ? 2 : 0),
                                    (const float2*)e->hue_lut.p, e->stream);
    if (code == hipSuccess) code = hipMemcpyAsync(out, o.p, n * sizeof(float), hipMemcpyDeviceToHost, e->stream);
    if (code == hipSuccess) code = sync_stream(e);
    release(a); release(b); release(o);
    if (code != hipSuccess) return hip_fail(e, code, "Error during filter difference kernel", __FILE__, __LINE__);
    return MM_OK;
}

}  // extern "C"
