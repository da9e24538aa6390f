/*
 * mosaic_hip.h -- C ABI of the MI355X photomosaic best-fit engine.
 *
 * Drop-in boundary for ONE path of MorganGrundy/MosaicMagnifique: what
 * CUDAPhotomosaicGenerator::generateBestFits() does on the device
 * (src/Photomosaic/CUDA/CUDAPhotomosaicGenerator.cpp:40-368) and
 * CPUPhotomosaicGenerator::generateBestFits() does on the host
 * (src/Photomosaic/CPUPhotomosaicGenerator.cpp:33-113): for every valid grid
 * cell, masked colour difference against every library image, plus the
 * neighbour repeat penalty, argmin -> library index.
 *
 * The reference has no FFI layer; its seam is the C++ virtual class
 * PhotomosaicGeneratorBase (src/Photomosaic/PhotomosaicGeneratorBase.h:32-112)
 * and the kernel wrappers of CUDA/PhotomosaicGenerator.cuh:6-43.  A reference
 * maintainer binds this header from a ~100-line HIPPhotomosaicGenerator
 * subclass; the Qt/OpenCV host, Grid/CellShape/ImageLibrary
 * plumbing and the padded best-fit grid layout stay as they are.
 *
 * Plain C: pointers + sizes, no C++/torch types.  All input pointers are HOST
 * memory unless a parameter says "device".  Every entry point returns an
 * mm_status; no exception crosses the boundary.  One call at a time per
 * engine; engines are independent (one per GPU / host thread).
 */
#ifndef MOSAIC_HIP_H
#define MOSAIC_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MM_ABI_VERSION 4

/* Return codes.  The reference returns bool and reports through a modal
 * CUDAErrMessageBox that separates "expected" errors (out of memory, bad
 * device) from unexpected ones (CUDA/CUDAUtility.h:32-62); the categories
 * below carry the same split, the text is in mm_last_error(). */
typedef enum mm_status {
    MM_OK = 0,
    MM_CANCELLED = 1,        /* cancel flag seen: generateBestFits() returns false */
    MM_INVALID_ARGUMENT = 2, /* ColourDifference::getFunction throws std::invalid_argument (ColourDifference.cpp:23) */
    MM_OUT_OF_MEMORY = 3,    /* "expected": pre-flight check of CUDAPhotomosaicGenerator.cpp:398-430 */
    MM_DEVICE_ERROR = 4,     /* "unexpected": any other HIP error, with name + file:line */
    MM_NOT_READY = 5         /* library / masks / target not staged yet */
} mm_status;

/* Colour difference selector = ColourDifference::Type
 * (src/Photomosaic/ColourDifference.h:13-19). */
enum { MM_RGB_EUCLIDEAN = 0, MM_CIE76 = 1, MM_CIEDE2000 = 2 };

/* GridUtility::PAD_GRID (src/Grid/GridUtility.h:33). */
#define MM_PAD_GRID 2

typedef struct mm_engine mm_engine;

/* One valid grid cell, as the host computes it per cell today:
 *  x, y              top-left of GridUtility::getRectAt (Grid/GridUtility.cpp:86-115),
 *                    target-pixel coordinates, may be negative / past the edge;
 *  row_min..col_max  the detail-space local bound returned by getCellAt
 *                    (PhotomosaicGeneratorBase.cpp:319-326), half open, in the
 *                    order of the CUDA path's targetArea[4]
 *                    (CUDAPhotomosaicGenerator.cpp:193-196);
 *  mask_id           flipH + 2*flipV, the index CUDAPhotomosaicGenerator.cpp:208 uses;
 *  grid_x, grid_y    padded grid position (x + PAD_GRID, y + PAD_GRID).
 * Cells must be passed in the reference's raster order (y outer, x inner):
 * the repeat penalty depends on it. */
typedef struct mm_cell {
    int32_t x, y;
    int32_t row_min, row_max, col_min, col_max;
    int32_t mask_id;
    int32_t grid_x, grid_y;
} mm_cell;

/* Per-call device timings (HIP events on the engine's stream), milliseconds.
 * Phase names follow the reference's TimingLogger ids
 * (CUDAPhotomosaicGenerator.cpp: cudaMemcpy, DiffReduce, Repeats, FindLowest). */
typedef struct mm_timing {
    double stage_ms;      /* H2D of cell records + cell gather ("cudaMemcpy") */
    double score_ms;      /* fused difference + reduction kernels ("DiffReduce") */
    double select_ms;     /* per-cell candidate selection (top-K) */
    double resolve_ms;    /* repeat penalty + argmin + exact tie-break ("Repeats"+"FindLowest") */
    double total_ms;      /* first launch to results on the host */
    int64_t score_launches;   /* launches of the dominant score kernel */
    int64_t pair_evals;       /* colour differences evaluated by it (active pixels x candidates) */
    int64_t exact_rescored;   /* candidates re-summed in reference order to break a near-tie */
    int64_t dense_rows;       /* cells whose candidate list overflowed (whole row recomputed) */
    double rescore_ms;        /* fp64 re-score of the filter's contenders (+ fp64-only re-runs) */
    int64_t dense_cells;      /* cells the fp32 filter could not vouch for: re-run fp64-only */
    int64_t edge_pairs;       /* pixel pairs on the CIEDE2000 hue discontinuity patched in fp64 */
    double seed_ms;           /* seeds + pruning limits ahead of the filter pass */
    double build_ms;          /* mm_build_photomosaic: owner map + compose kernels */
    int64_t spec_rounds;      /* resolve passes of the speculate-and-verify loop (1 = no cell needed deepening) */
    int64_t deepened_cells;   /* cells whose short candidate list had to be extended to the full length */
    int64_t resolve_sweeps;   /* parallel sweeps the repeat-penalty resolve took to reach its fix point */
    /* the next three: the filter kernels' always-on work counters (0 when the call ran without the early exit) */
    int64_t visited_groups;   /* 256-pixel groups the filter kernel evaluated (always-on counter) */
    int64_t total_groups;     /* groups a full evaluation of every (cell, candidate) would take */
    int64_t bound_dropped;    /* candidates the block bound dropped before any pixel was read */
    int64_t band_alarms;      /* sampled cells where a candidate the fp32 band excluded beat a contender in fp64
                               * (must stay 0; such a cell is re-done fp64-only) */
    int64_t filter_tile;      /* candidates per work item the last filter launch ran with (see "tile" / "item_auto") */
    int64_t host_syncs;       /* times the host waited for the device during the call */
    int64_t fast_path;        /* 1: the whole generate was ONE submission read back in one result block (option
                               * "fast_path"); 0: the step-by-step path ran (or finished the call) */
    double preprocess_ms;     /* mm_set_library_u8[_sized] / mm_set_target_u8: the resize + colour conversion kernels */
    int64_t filter_off_range; /* 1: the f32 pixels handed over (mm_set_library / mm_set_target) leave the box the fp32 filter's
                               * error band was validated on -- Lab: channel 0 in [-0.5, 100.5], chroma <= 150; RGB / CIE76:
                               * |channel| <= 256; or hold a non-finite value -- so the call ran fp64-only (same result) */
    int64_t pretier;          /* 1: the last CIEDE2000 filter launch ran with the packed-fp16 pre-tier (option "pretier") */
    double audit_ms;          /* the audit of the candidates the CIEDE2000 filter dropped (option "audit"), between the filter and the selection */
    int64_t voided_launches;  /* filter launches whose scores were thrown away -- the pre-tier's canary tripped, or the audit found a dropped
                               * candidate below the limit -- and whose cells were re-done fp64-only (must stay 0; same result either way) */
} mm_timing;

/* What a FIRST call pays besides the hot path, as host wall clocks accumulated since mm_create: the reference builds
 * a fresh generator per run (MainWindow.cpp:584-607, Benchmark/Benchmark_Generator.h:17-104) and allocates / uploads
 * everything inside generateBestFits (CUDAPhotomosaicGenerator.cpp:82,372-510), so for its everyday job sizes these
 * -- not the kernels -- are the time a caller sees. */
typedef struct mm_stage_timing {
    double create_ms;         /* mm_create: stream, events (the HIP context is the process's) */
    double alloc_ms;          /* device allocations (hipMalloc), all buffers of the engine */
    double pinned_ms;         /* page-locked host buffers */
    double h2d_ms;            /* host -> device copies of library, masks, target (mm_set_*; u8 inputs: the copy alone) */
    double preprocess_ms;     /* u8 -> f32 / Lab conversion and resize kernels (mm_set_*_u8) */
    double planes_ms;         /* library planes, interleaved copy, block statistics, means (once per library / mask change) */
    int64_t allocations;      /* device allocations made */
    int64_t allocated_bytes;
    int64_t reused_engine;    /* 1: mm_create adopted the parked resources of a destroyed engine of this process */
} mm_stage_timing;

/* Progress: called on the calling thread between device batches with
 * monotone values up to the number of cells (the reference emits one
 * progress(int) per cell, CPUPhotomosaicGenerator.cpp:87-88).  cancel is
 * polled at the same points (PhotomosaicGeneratorBase::cancel, :217-220). */
typedef void (*mm_progress_cb)(int cells_done, void *user);

/* ---- lifetime ---------------------------------------------------------- */

int mm_abi_version(void);
/* Number of visible HIP devices (MainWindow.cpp:162-214 enumerates them). */
mm_status mm_device_count(int *count);
/* CUDAPhotomosaicGenerator(const int device), CUDA/CUDAPhotomosaicGenerator.h:31. */
mm_status mm_create(int device, mm_engine **out);
/* Destroys the engine.  Its DEVICE RESOURCES (stream, events, device buffers, the page-locked result mirror) are
 * parked -- one set per device -- and adopted by the next mm_create on that device, so that a fresh generator per
 * run (MainWindow.cpp:584-607) does not pay the allocations again (mm_stage_timing).  Nothing logical survives:
 * a new engine has no library, masks or target and the default options.  A parked set holds at most 5 % of the
 * device memory that would be free with the set released (asked at mm_destroy: ~14 GB on an idle MI355X, so a
 * 4096-cell x 5000-image generator's 2.4 GB of workspaces stay and its successor allocates nothing); the largest
 * buffers are released first when the set does not fit.  MOSAIC_HIP_KEEP_MB (environment) pins a fixed cap in
 * megabytes instead.  Measured by the driver in round 5 with a fixed 256 MB cap: 73-76 ms of hipMalloc per fresh
 * config-D generator -- allocation of multi-GB buffers is NOT sub-millisecond on every box. */
void mm_destroy(mm_engine *e);
/* Releases the parked resources of destroyed engines (all devices); returns how many sets were released.
 * MOSAIC_HIP_KEEP=0 in the environment turns the parking off altogether (mm_destroy then frees everything, as the
 * reference does before generateBestFits returns, CUDAPhotomosaicGenerator.cpp:356). */
int mm_trim_cache(void);
/* Text of the last failure on this engine (e == NULL: last mm_create failure):
 * "<what>: <hipErrorName> (<code>) at <file>:<line>", the shape of
 * createCUDAErrStr (CUDA/CUDAUtility.h:27-30). */
const char *mm_last_error(const mm_engine *e);

/* ---- staging (once; replaces the per-generate cudaMalloc + H2D of
 * CUDAPhotomosaicGenerator.cpp:113-149,372-510) -------------------------- */

/* Library after preprocessLibraryImages (PhotomosaicGeneratorBase.cpp:255-290):
 * n_images square images, f32 HWC [n][size][size][3], BGR in [0,255] for
 * RGB_EUCLIDEAN or Lab for CIE76/CIEDE2000.  Copied; the caller may free. */
mm_status mm_set_library(mm_engine *e, const float *images, int n_images, int size);
/* The 4 detail-cell masks, u8 [4][size][size], order (flipH,flipV) =
 * (0,0),(1,0),(0,1),(1,1) (CUDAPhotomosaicGenerator.cpp:122-134); != 0 = active. */
mm_status mm_set_masks(mm_engine *e, const uint8_t *masks, int size);
/* Main image after preprocessMainImage (PhotomosaicGeneratorBase.cpp:223-252),
 * f32 HWC [rows][cols][3].  Cells are gathered from it on the device. */
mm_status mm_set_target(mm_engine *e, const float *image, int rows, int cols);

/* ---- pre-processing and detail / size steps on the device ------------------
 * (SURVEY 8f N2 / N3).  OpenCV is not part of the reference checkout, so the
 * arithmetic of these steps is SPECIFIED in csrc/kernels.hpp (cell_pixel,
 * preprocess_u8_kernel) and restated by the CPU checker of the test suite; only integer resize
 * factors (detail 100/n %, even cell sizes per size step) are covered. */

/* preprocessLibraryImages() (PhotomosaicGeneratorBase.cpp:255-290): u8 BGR
 * [n_images][size][size][3] -> INTER_AREA shrink by `shrink` on the u8 images
 * (detail cell size = size / shrink; 1 = none) -> f32 BGR (MM_RGB_EUCLIDEAN) or
 * f32 Lab (MM_CIE76 / MM_CIEDE2000).  Replaces mm_set_library. */
mm_status mm_set_library_u8(mm_engine *e, const uint8_t *bgr, int n_images, int size,
                            int shrink, int diff_type);
/* The same for ANY detail cell size <= size (a detail level like 75 %): where size / detail_size
 * is not whole the shrink is cv::resize's general INTER_AREA (computeResizeAreaTab +
 * ResizeArea_Invoker, f32 accumulators, table order; csrc/kernels.hpp resize_area_u8_kernel). */
mm_status mm_set_library_u8_sized(mm_engine *e, const uint8_t *bgr, int n_images, int size,
                                  int detail_size, int diff_type);

/* preprocessMainImage() (PhotomosaicGeneratorBase.cpp:223-252), colour scheme
 * NONE: u8 BGR [rows][cols][3] -> f32 BGR or Lab.  Replaces mm_set_target. */
mm_status mm_set_target_u8(mm_engine *e, const uint8_t *bgr, int rows, int cols, int diff_type);

/* One variant of ColourScheme::getColourScheme{Complementary,Triadic,Compound,Tetradic,
 * Analagous} (ColourScheme.cpp:44-177): the u8 BGR image with its hue rotated by `degrees`
 * (u8 -> f32 -> BGR2HSV_FULL -> H = fmod(H + degrees, 360) -> HSV2BGR_FULL -> u8), written to
 * the caller's `out` [rows][cols][3].  The host builds the variant list of a scheme from it
 * (the rotations are 180 | 120,240 | 150,210 | 90,180,270 | 30,60,90) and stages the variant it
 * compares against with mm_set_target_u8.  OpenCV's cvtColor is a third-party dependency absent
 * from the reference checkout: the arithmetic is specified in csrc/kernels.hpp (hue_rotate_px). */
mm_status mm_hue_rotate_u8(mm_engine *e, const uint8_t *bgr, int rows, int cols, float degrees,
                           uint8_t *out);

/* ImageUtility::batchResizeMat(lib) between size steps
 * (CPUPhotomosaicGenerator.cpp:95-99): the resident f32 library is replaced by
 * its INTER_AREA reduction by `factor` (2 per size step). */
mm_status mm_shrink_library(mm_engine *e, int factor);

/* Detail level 1/factor (getCellAt's resize of the cell to the detail size,
 * PhotomosaicGeneratorBase.cpp:313-315): cell records keep x, y in image
 * pixels of the NORMAL cell, bounds and masks are in detail pixels, and each
 * detail pixel of a cell is the mean of a factor x factor block of the
 * zero-filled normal cell.  1 (default) = detail 100 %. */
mm_status mm_set_cell_scale(mm_engine *e, int factor);
/* Any detail level: the normal cell of normal_size pixels is resized to detail_size (= the
 * library's current size).  A whole ratio is mm_set_cell_scale(normal_size / detail_size); otherwise
 * each detail pixel is cv::resize's general INTER_AREA of the zero-filled normal cell
 * (computeResizeAreaTab + ResizeArea_Invoker<float, float>: f32, table order, no fused
 * multiply-adds; csrc/kernels.hpp cell_pixel). */
mm_status mm_set_cell_resize(mm_engine *e, int normal_size, int detail_size);

/* Read-back of the resident (pre-processed) inputs, for tests and for feeding a
 * CPU generator the same arrays: library f32 [n_images][size][size][3] at its
 * current size (mm_library_shape), main image f32 [rows][cols][3]. */
mm_status mm_library_shape(mm_engine *e, int *n_images, int *size);
mm_status mm_get_library(mm_engine *e, float *images_out);
mm_status mm_get_target(mm_engine *e, float *image_out);

/* ---- the hot path ------------------------------------------------------ */

/* generateBestFits() for one size step.
 *  best_fit_out   [grid_h*grid_w] row-major padded grid, the layout of the CUDA
 *                 path's bestFit[] (cellPosition = grid_y*grid_w + grid_x);
 *                 entries not named by `cells` hold the sentinel n_images
 *                 (CUDAPhotomosaicGenerator.cpp:140-149,286,320);
 *  best_score_out optional [grid_h*grid_w], bestVariant per cell (NaN elsewhere).
 * Returns MM_CANCELLED if *cancel became non-zero. */
mm_status mm_generate(mm_engine *e, const mm_cell *cells, int n_cells,
                      int grid_w, int grid_h, int diff_type,
                      int repeat_range, int repeat_addition,
                      int64_t *best_fit_out, double *best_score_out,
                      mm_progress_cb progress, void *progress_user,
                      volatile int *cancel);

/* Split form for the cell-sharded multi-GPU run: each rank scores its own
 * contiguous block of cells, the per-cell candidate lists are all-gathered,
 * every rank then resolves the whole grid identically.
 * mm_candidates_per_cell: how many (index, score) candidates per cell the
 * resolve needs: the window of calculateRepeats (CPUPhotomosaicGenerator.cpp:
 * 185-225) holds at most R(2R+1)+R neighbours, so the winner is among the
 * R(2R+1)+R+1 lowest base scores; +1 runner-up for the near-tie check. */
int mm_candidates_per_cell(int n_images, int repeat_range, int repeat_addition);
/* cand_index_out [n_cells][k] int32, cand_score_out [n_cells][k] f64: the k
 * lowest base scores per cell ascending by (score, index); index -1 = no such
 * candidate.  A LAST index <= -2 encodes -(index + 2) and marks a list that was
 * cut through a cluster of near-tied scores: mm_resolve then recomputes that
 * cell's whole row, so the lists can be all-gathered as they are.  Host pointers, or
 * device pointers if outputs_on_device != 0 (results complete on return). */
mm_status mm_score_cells(mm_engine *e, const mm_cell *cells, int n_cells,
                         int diff_type, int k,
                         int32_t *cand_index_out, double *cand_score_out,
                         int outputs_on_device,
                         mm_progress_cb progress, void *progress_user,
                         volatile int *cancel);
/* Speculative short lists.  With a repeat penalty, mm_candidates_per_cell is a worst-case
 * bound that is almost never needed.  mm_speculative_candidates gives the length K_s of the
 * short lists worth scoring first (= the full length when speculation does not pay; spec_k
 * 0 = off, 10 = the engine's default).  mm_score_cells_partial fills only the first k_fill
 * entries of k-wide rows.  mm_resolve_verify resolves rows of which deep[c] != 0 hold the
 * full length and the others k_fill entries, and reports in insufficient_out[c] the cells
 * whose pick it could not prove right (an unlisted candidate has a base score >= the list's
 * last one and a penalty >= 0): re-score those with the full length, mark them deep, resolve
 * again.  mm_generate runs this loop itself. */
int mm_speculative_candidates(int n_images, int repeat_range, int repeat_addition, int spec_k);
mm_status mm_score_cells_partial(mm_engine *e, const mm_cell *cells, int n_cells, int diff_type, int k, int k_fill,
                                 int32_t *cand_index_out, double *cand_score_out, int outputs_on_device,
                                 mm_progress_cb progress, void *progress_user, volatile int *cancel);
mm_status mm_resolve_verify(mm_engine *e, const mm_cell *cells, int n_cells, int grid_w, int grid_h, int diff_type,
                            int repeat_range, int repeat_addition, int k, int k_fill,
                            const int32_t *cand_index, const double *cand_score, int inputs_on_device,
                            const uint8_t *deep, uint8_t *insufficient_out,
                            int64_t *best_fit_out, double *best_score_out);

/* cells = ALL valid cells of the grid in raster order with their candidate lists. */
mm_status mm_resolve(mm_engine *e, const mm_cell *cells, int n_cells,
                     int grid_w, int grid_h, int diff_type,
                     int repeat_range, int repeat_addition, int k,
                     const int32_t *cand_index, const double *cand_score,
                     int inputs_on_device,
                     int64_t *best_fit_out, double *best_score_out);

/* ---- composing the result -------------------------------- */

/* The library as setLibrary() received it: u8 BGR [n][size][size][3], n and order as
 * in mm_set_library.  Needed only by mm_build_photomosaic. */
mm_status mm_set_library_bgr(mm_engine *e, const uint8_t *images, int n_images, int size);
/* PhotomosaicGeneratorBase::buildPhotomosaic (PhotomosaicGeneratorBase.cpp:110-207) for one
 * size step: BGRA canvas of `background`, every cell with a value painted in raster order
 * through its flipped mask (later cells win where shapes overlap), library alpha 255.
 *  best_fit     [grid_h*grid_w] as mm_generate wrote it (entries < 0 or >= n_images: no value)
 *  mosaic_out   u8 BGRA [rows][cols][4] (host) */
mm_status mm_build_photomosaic(mm_engine *e, const mm_cell *cells, int n_cells, int grid_w, int grid_h,
                               const int64_t *best_fit, const uint8_t background[4], int rows, int cols,
                               uint8_t *mosaic_out);
/* The same for size step `step_index` of several (:123-202): step 0 starts a new mosaic;
 * a later step paints only the pixels no earlier step's cell masks cover (:189-196).
 * Between the steps the caller stages the step's NORMAL masks (mm_set_masks) and halves the
 * u8 library (mm_shrink_library_bgr, ImageUtility::batchResizeMat, :126-127).  The engine
 * keeps the mosaic; mosaic_out may be NULL for all but the last step. */
mm_status mm_build_photomosaic_step(mm_engine *e, const mm_cell *cells, int n_cells, int grid_w, int grid_h,
                                    const int64_t *best_fit, const uint8_t background[4], int rows, int cols,
                                    int step_index, uint8_t *mosaic_out);
mm_status mm_shrink_library_bgr(mm_engine *e, int factor);

/* ---- the entropy subdivision of the grid ------------------------
 * GridGenerator::getGridState hands a cell of a size step that is not the last to the next (half-size) step when
 * the grey-level entropy of the image under its bounded, flipped detail mask reaches 0.7 * MAX_ENTROPY
 * (Grid/GridGenerator.cpp:143-189, Other/ImageUtility.cpp:189-242).  The host keeps the integer geometry and the
 * bound merging (Grid/GridBounds.cpp); the per-cell image work -- resize of the visible part to the mask's size,
 * BGR2GRAY, masked 256-bin histogram, entropy -- runs here, one workgroup per cell.
 * mm_set_main_image_bgr: the main image as setMainImage receives it (u8 BGR [rows][cols][3]); kept on the device.
 * mm_cell_entropy: cells = the step's cells inside an active bound, records as for mm_generate (x, y = rect origin
 * of the NORMAL cell of normal_size pixels, the bound in DETAIL pixels, mask_id); the staged masks (mm_set_masks)
 * must be the step's detail masks.  exceeded_out[n]: 1 = entropy >= 5.6 (split the cell); entropy_out optional. */
mm_status mm_set_main_image_bgr(mm_engine *e, const uint8_t *bgr, int rows, int cols);
mm_status mm_cell_entropy(mm_engine *e, const mm_cell *cells, int n_cells, int normal_size,
                          uint8_t *exceeded_out, double *entropy_out);

/* ---- one process, several GPUs ---------------------------------
 * The reference drives one device (CUDAPhotomosaicGenerator(const int device), CUDA/CUDAPhotomosaicGenerator.h:31;
 * the device list it chooses from: MainWindow.cpp:162-214).  A group holds one engine per listed device: the valid
 * cells (raster order) are cut into contiguous shards of ceil(n / devices), inputs are replicated, every device
 * scores its shard, ONE RCCL all-gather (ncclAllGather inside ncclGroupStart / End, one communicator per device
 * from ncclCommInitAll) hands every device every candidate list, and every device runs the same verifying resolve;
 * unverified speculative picks are deepened by their owners and gathered again.  The grid equals mm_generate's on
 * one device, bit for bit.  RCCL refuses a device listed twice (what a one-GPU box can rehearse): such a group, or
 * one whose communicator cannot be built, gathers through host memory instead -- same buffers, same resolve, same
 * result (mm_group_uses_rccl).  Host threads: one per device inside the calls below; the calls themselves are
 * blocking and not re-entrant per group. */
typedef struct mm_group mm_group;
mm_status mm_group_create(const int *devices, int n_devices, mm_group **out);
void mm_group_destroy(mm_group *g);
/* g == NULL: the last mm_group_create failure. */
const char *mm_group_last_error(const mm_group *g);
int mm_group_size(const mm_group *g);
/* 1: the candidate lists travel by ncclAllGather; 0: through host memory (see above). */
int mm_group_uses_rccl(const mm_group *g);
/* The engine on devices[index] (owned by the group): for mm_get_timing, mm_set_cell_resize, mm_shrink_library,
 * mm_build_photomosaic ... on one device. */
mm_engine *mm_group_engine(mm_group *g, int index);
/* Replicated staging: mm_set_library / mm_set_library_u8_sized / mm_set_masks / mm_set_target / mm_set_target_u8 /
 * mm_set_option on every engine of the group (one host thread per device). */
mm_status mm_group_set_library(mm_group *g, const float *images, int n_images, int size);
mm_status mm_group_set_library_u8(mm_group *g, const uint8_t *bgr, int n_images, int size, int detail_size, int diff_type);
mm_status mm_group_set_masks(mm_group *g, const uint8_t *masks, int size);
mm_status mm_group_set_target(mm_group *g, const float *image, int rows, int cols);
mm_status mm_group_set_target_u8(mm_group *g, const uint8_t *bgr, int rows, int cols, int diff_type);
mm_status mm_group_set_option(mm_group *g, const char *name, int value);
/* generateBestFits() for one size step over the group's devices; arguments and outputs as mm_generate. */
mm_status mm_group_generate(mm_group *g, const mm_cell *cells, int n_cells, int grid_w, int grid_h, int diff_type,
                            int repeat_range, int repeat_addition, int64_t *best_fit_out, double *best_score_out,
                            mm_progress_cb progress, void *progress_user, volatile int *cancel);

/* Engine options.  Three switches belong to the boundary; every other name is tuning or diagnostics and lives in
 * mosaic_hip_tuning.h (results are identical whatever the options: tests/test_gpu_fuzz.py).
 *   "filter"    (default 1)  two-tier scoring: an fp32 filter pass over every (cell, candidate), then the fp64 score of
 *                            each cell's contenders (near-ties re-summed in the reference's order with its own
 *                            arithmetic); 0 scores every pixel pair in fp64 -- the escape hatch.
 *   "prune"     (default 1)  the filter pass abandons a candidate once its running sum passes a per-cell upper bound
 *                            (the device form of the reference's early exit, CPUPhotomosaicGenerator.cpp:149-161).
 *   "max_batch" (default 0)  cap on the cells scored per device batch; 0 = sized from free device memory.
 * Unknown names return MM_INVALID_ARGUMENT. */
mm_status mm_set_option(mm_engine *e, const char *name, int value);

/* Timings of the last mm_generate / mm_score_cells (+ mm_resolve) on e. */
mm_status mm_get_timing(const mm_engine *e, mm_timing *out);
/* The stage clocks of e since mm_create (see mm_stage_timing). */
mm_status mm_get_stage_timing(const mm_engine *e, mm_stage_timing *out);

/* ---- per-pair colour difference ---------------------------------------- */

/* n pairs of f32 triplets -> fp64 differences on the device: the counterpart
 * of ColourDifference::getCUDAFunction applied to a flat image
 * (ColourDifference.cpp:164-172; exercised by test/tst_ColourDifference.h:233-466).
 * first = main-image pixel, second = library pixel. */
mm_status mm_colour_diff_batch(mm_engine *e, int diff_type,
                               const float *first, const float *second,
                               size_t n, double *out);

#ifdef __cplusplus
}
#endif
#endif /* MOSAIC_HIP_H */
