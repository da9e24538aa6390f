/*
 * oracle.c -- CPU restatement of the MosaicMagnifique best-fit path.
 * TEST INFRASTRUCTURE ONLY (see oracle.h).  Plain C99, fp64, one thread,
 * built with -ffp-contract=off so no product is fused: the reference is an
 * MSVC x64 build (SSE2, no FMA contraction).
 *
 * Operation ORDER matters here: per-pixel terms are summed sequentially in
 * fp64 and compared with strict <, so every expression below keeps the
 * association of the reference expression it cites.
 */
#include "oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

/* ------------------------------------------------------------------------ */
/* Colour difference                                                         */
/* ------------------------------------------------------------------------ */

/* ColourDifference::calculateRGBEuclidean, Photomosaic/ColourDifference.cpp:28-33.
 * CIE76 is the same function on Lab triplets (ColourDifference.h:41-43). */
double orc_rgb_euclidean(const double p[3], const double q[3])
{
    const double d0 = p[0] - q[0], d1 = p[1] - q[1], d2 = p[2] - q[2];
    return sqrt(d0 * d0 + d1 * d1 + d2 * d2);
}

/* ColourDifference::degToRad, ColourDifference.cpp:36-39: (deg * pi) / 180. */
static double rad(double deg) { return (deg * M_PI) / 180.0; }

/* x^7 by six left-to-right multiplies, as ColourDifference.cpp:52 and :135. */
static double seventh_power(double x) { return x * x * x * x * x * x * x; }

/* Hue angle in [0, 2pi), ColourDifference.cpp:61-83. */
static double hue_angle(double b, double a_prime, double two_pi)
{
    double h;
    if (b == 0.0 && a_prime == 0.0)
        return 0.0;
    h = atan2(b, a_prime);
    if (h < 0.0)
        h += two_pi;
    return h;
}

/* ColourDifference::calculateCIEDE2000, ColourDifference.cpp:42-158.
 * Sharma-2005 formulation in radians, kL = kC = kH = 1.  `p` is the cell
 * (main image) pixel, `q` the library pixel: the call site passes them in
 * that order (CPUPhotomosaicGenerator.cpp:159). */
double orc_ciede2000(const double p[3], const double q[3])
{
    const double two_pi = rad(360.0), pi = rad(180.0);
    const double twentyfive_7 = 6103515625.0;                        /* :46 */

    /* chroma and the G correction, :48-53 */
    const double c1 = sqrt((p[1] * p[1]) + (p[2] * p[2]));
    const double c2 = sqrt((q[1] * q[1]) + (q[2] * q[2]));
    const double c_bar = (c1 + c2) / 2.0;
    const double c_bar7 = seventh_power(c_bar);
    const double g = 0.5 * (1.0 - sqrt(c_bar7 / (c_bar7 + twentyfive_7)));

    /* a', C', h' for both colours, :55-83 */
    const double a1p = (1.0 + g) * p[1];
    const double a2p = (1.0 + g) * q[1];
    const double c1p = sqrt((a1p * a1p) + (p[2] * p[2]));
    const double c2p = sqrt((a2p * a2p) + (q[2] * q[2]));
    const double h1p = hue_angle(p[2], a1p, two_pi);
    const double h2p = hue_angle(q[2], a2p, two_pi);

    /* deltas, :85-102 */
    const double d_l = q[0] - p[0];
    const double d_c = c2p - c1p;
    const double cp_prod = c1p * c2p;
    double d_h_small = 0.0;
    if (cp_prod != 0.0) {
        d_h_small = h2p - h1p;
        if (d_h_small < -pi)
            d_h_small += two_pi;
        else if (d_h_small > pi)
            d_h_small -= two_pi;
    }
    const double d_h = 2.0 * sqrt(cp_prod) * sin(d_h_small / 2.0);

    /* means, :104-122 */
    const double l_bar = (p[0] + q[0]) / 2.0;
    const double cp_bar = (c1p + c2p) / 2.0;
    const double h_sum = h1p + h2p;
    double h_bar;
    if (c1p * c2p == 0.0)
        h_bar = h_sum;
    else if (fabs(h1p - h2p) <= pi)
        h_bar = h_sum / 2.0;
    else if (h_sum < two_pi)
        h_bar = (h_sum + two_pi) / 2.0;
    else
        h_bar = (h_sum - two_pi) / 2.0;

    /* T, :124-127 */
    const double t = 1.0 - (0.17 * cos(h_bar - rad(30.0)))
                   + (0.24 * cos(2.0 * h_bar))
                   + (0.32 * cos((3.0 * h_bar) + rad(6.0)))
                   - (0.20 * cos((4.0 * h_bar) - rad(63.0)));

    /* delta-theta with the expanded ((h-275)/25)^2, :132-133 */
    const double d_theta = rad(30.0) *
        exp(-((h_bar * (h_bar - rad(550.0)) + (rad(275.0) * rad(275.0)))
              / (rad(25.0) * rad(25.0))));

    /* R_C, S_L (expanded (L-50)^2), S_C, S_H, R_T, :135-145 */
    const double cp_bar7 = seventh_power(cp_bar);
    const double r_c = 2.0 * sqrt(cp_bar7 / (cp_bar7 + twentyfive_7));
    const double s_l = 1.0 + ((0.015 * (l_bar * (l_bar - 100.0) + 2500.0))
                              / sqrt(20 + l_bar * (l_bar - 100.0) + 2500.0));
    const double s_c = 1.0 + (0.045 * cp_bar);
    const double s_h = 1.0 + (0.015 * cp_bar * t);
    const double r_t = (-sin(2.0 * d_theta)) * r_c;

    /* final quadratic form, :153-157 */
    return sqrt((d_l * d_l) / (s_l * s_l)
              + (d_c * d_c) / (s_c * s_c)
              + (d_h * d_h) / (s_h * s_h)
              + (r_t * (d_c / s_c) * (d_h / s_h)));
}

/* ColourDifference::getFunction, ColourDifference.cpp:16-25. */
double orc_colour_diff(int type, const double p[3], const double q[3])
{
    switch (type) {
    case ORC_RGB_EUCLIDEAN:
    case ORC_CIE76:
        return orc_rgb_euclidean(p, q);
    case ORC_CIEDE2000:
        return orc_ciede2000(p, q);
    default:
        return NAN;
    }
}

void orc_colour_diff_batch(int type, const float *first, const float *second,
                           size_t n, double *out)
{
    for (size_t i = 0; i < n; ++i) {
        const double p[3] = { first[3 * i], first[3 * i + 1], first[3 * i + 2] };
        const double q[3] = { second[3 * i], second[3 * i + 1], second[3 * i + 2] };
        out[i] = orc_colour_diff(type, p, q);
    }
}

/* ------------------------------------------------------------------------ */
/* Geometry                                                                  */
/* ------------------------------------------------------------------------ */

static int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

/* Cells needed along one axis, GridUtility::calculateGridSize,
 * Grid/GridUtility.cpp:25-52: pairwise when the alternate spacing differs. */
static int cells_along(int extent, int spacing, int alt_spacing)
{
    if (spacing != alt_spacing)
        return 2 * ((extent + spacing + alt_spacing - 1) / (spacing + alt_spacing));
    return (extent + spacing - 1) / spacing;
}

void orc_grid_size(const orc_shape *s, int img_w, int img_h, int pad,
                   int *grid_w, int *grid_h)
{
    *grid_w = cells_along(img_w, s->col_spacing, s->alt_col_spacing) + pad;
    *grid_h = cells_along(img_h, s->row_spacing, s->alt_row_spacing) + pad;
}

/* Start of cell n along one axis, GridUtility::getRectAt,
 * Grid/GridUtility.cpp:86-115.  C integer division truncates toward zero,
 * so for negative n the roles of spacing / alternate spacing swap (:95-99). */
static int axis_start(int n, int spacing, int alt_spacing)
{
    const int half = n / 2;
    const int rest = n - half;
    if (n < 0)
        return rest * spacing + half * alt_spacing;
    return half * spacing + rest * alt_spacing;
}

orc_rect orc_rect_at(const orc_shape *s, int x, int y)
{
    orc_rect r;
    r.x = axis_start(x, s->col_spacing, s->alt_col_spacing);
    r.x += (y % 2 != 0) ? s->alt_row_offset : 0;
    r.y = axis_start(y, s->row_spacing, s->alt_row_spacing);
    r.y += (x % 2 != 0) ? s->alt_col_offset : 0;
    r.w = s->size;
    r.h = s->size;
    return r;
}

/* GridUtility::getFlipStateAt, Grid/GridUtility.cpp:118-132. */
void orc_flip_state_at(const orc_shape *s, int x, int y, int *flip_h, int *flip_v)
{
    int h = 0, v = 0;
    if (s->alt_col_flip_h && x % 2 != 0) h = !h;
    if (s->alt_row_flip_h && y % 2 != 0) h = !h;
    if (s->alt_col_flip_v && x % 2 != 0) v = !v;
    if (s->alt_row_flip_v && y % 2 != 0) v = !v;
    *flip_h = h;
    *flip_v = v;
}

/* GridGenerator::getGridState + findCellState with sizeSteps == 0
 * (Grid/GridGenerator.cpp:29-110, :113-140, :191-192): the only bound is the
 * whole image, a cell is optional(0) iff its rect clamped to the image is
 * non-empty on both axes, and the entropy branch (:143) is never taken. */
void orc_grid_state(const orc_shape *s, int img_w, int img_h,
                    int grid_w, int grid_h, int64_t *state)
{
    for (int gy = 0; gy < grid_h; ++gy) {
        for (int gx = 0; gx < grid_w; ++gx) {
            const orc_rect r = orc_rect_at(s, gx - ORC_PAD_GRID, gy - ORC_PAD_GRID);
            const int y0 = clampi(r.y, 0, img_h), y1 = clampi(r.y + r.h, 0, img_h);
            const int x0 = clampi(r.x, 0, img_w), x1 = clampi(r.x + r.w, 0, img_w);
            state[(size_t)gy * grid_w + gx] = (y0 != y1 && x0 != x1) ? 0 : -1;
        }
    }
}

/* PhotomosaicGeneratorBase::getCellAt, PhotomosaicGeneratorBase.cpp:293-329,
 * at detail == 1 (resizeImage is the identity when the factor is 1,
 * Other/ImageUtility.cpp:46-47).  The int truncation and the min/max guards
 * of :319-326 are kept, with detail fixed at 1.0. */
orc_rect orc_cell_at(const orc_shape *s, int x, int y, const float *img,
                     int rows, int cols, float *cell_out)
{
    const orc_rect g = orc_rect_at(s, x, y);
    const int size = s->size;
    const int y0 = clampi(g.y, 0, rows), y1 = clampi(g.y + g.h, 0, rows);
    const int x0 = clampi(g.x, 0, cols), x1 = clampi(g.x + g.w, 0, cols);
    const orc_rect local = { x0 - g.x, y0 - g.y, x1 - x0, y1 - y0 };

    memset(cell_out, 0, sizeof(float) * 3u * (size_t)size * (size_t)size);
    for (int r = 0; r < local.h; ++r) {
        const float *src = img + ((size_t)(y0 + r) * cols + x0) * 3u;
        float *dst = cell_out + ((size_t)(local.y + r) * size + local.x) * 3u;
        memcpy(dst, src, sizeof(float) * 3u * (size_t)local.w);
    }

    const double detail = 1.0;
    orc_rect b;
    b.x = (int)(local.x * detail); if (b.x > size - 1) b.x = size - 1;
    b.y = (int)(local.y * detail); if (b.y > size - 1) b.y = size - 1;
    b.w = (int)(local.w * detail); if (b.w < 1) b.w = 1;
    b.h = (int)(local.h * detail); if (b.h < 1) b.h = 1;
    if (b.w > size - b.x) b.w = size - b.x;
    if (b.h > size - b.y) b.h = size - b.y;
    return b;
}

/* ------------------------------------------------------------------------ */
/* Repeats, best fit, driver                                                 */
/* ------------------------------------------------------------------------ */

static void bump(int64_t key, int addition, int64_t *keys, int *vals, int *n, int cap)
{
    for (int i = 0; i < *n; ++i)
        if (keys[i] == key) { vals[i] += addition; return; }
    if (*n < cap) { keys[*n] = key; vals[*n] = addition; ++*n; }
}

/* CPUPhotomosaicGenerator::calculateRepeats, CPUPhotomosaicGenerator.cpp:185-225.
 * Rows clamp(y-R,0,H) .. y-1 over columns clamp(x-R,0,W) .. clamp(x+R,0,W-1)
 * inclusive, then the same row's columns startX .. x-1; nullopt skipped. */
int orc_repeats(const int64_t *grid, int grid_w, int grid_h, int px, int py,
                int range, int addition, int64_t *keys, int *vals, int cap)
{
    int n = 0;
    const int start_y = clampi(py - range, 0, grid_h);
    const int start_x = clampi(px - range, 0, grid_w);
    const int end_x = clampi(px + range, 0, grid_w - 1);

    for (int ry = start_y; ry < py; ++ry)
        for (int rx = start_x; rx <= end_x; ++rx) {
            const int64_t c = grid[(size_t)ry * grid_w + rx];
            if (c >= 0) bump(c, addition, keys, vals, &n, cap);
        }
    for (int rx = start_x; rx < px; ++rx) {
        const int64_t c = grid[(size_t)py * grid_w + rx];
        if (c >= 0) bump(c, addition, keys, vals, &n, cap);
    }
    return n;
}

/* CPUPhotomosaicGenerator::findCellBestFit, CPUPhotomosaicGenerator.cpp:117-181
 * (one main-image variant).  Library index ascending; variant starts at the
 * repeat value as int -> double (:143-144); rows then columns of the local
 * bound, row-major, skipping mask == 0 (:149-161); strict < keeps the lowest
 * index on ties (:165-169). */
int64_t orc_find_cell_best_fit(int type, const float *cell, orc_rect bounds,
                               const uint8_t *mask, int size,
                               const float *lib, int64_t n_lib,
                               const int64_t *rep_keys, const int *rep_vals, int n_rep,
                               int early_exit, double *best_out)
{
    int64_t best_fit = -1;
    double best = 1.7976931348623157e308; /* std::numeric_limits<double>::max() */
    const size_t img_stride = (size_t)size * size * 3u;
    const int row_end = bounds.y + bounds.h, col_end = bounds.x + bounds.w;

    for (int64_t i = 0; i < n_lib; ++i) {
        double variant = 0;
        for (int k = 0; k < n_rep; ++k)
            if (rep_keys[k] == i) { variant = rep_vals[k]; break; }

        const float *im = lib + (size_t)i * img_stride;
        for (int row = bounds.y; row < row_end && (!early_exit || variant < best); ++row) {
            const float *p_main = cell + (size_t)row * size * 3u;
            const float *p_im = im + (size_t)row * size * 3u;
            const uint8_t *p_mask = mask + (size_t)row * size;
            for (int col = bounds.x; col < col_end && (!early_exit || variant < best); ++col) {
                if (p_mask[col] != 0) {
                    const double a[3] = { p_main[3 * col], p_main[3 * col + 1], p_main[3 * col + 2] };
                    const double b[3] = { p_im[3 * col], p_im[3 * col + 1], p_im[3 * col + 2] };
                    variant += (type == ORC_CIEDE2000) ? orc_ciede2000(a, b)
                                                       : orc_rgb_euclidean(a, b);
                }
            }
        }
        if (variant < best) {
            best = variant;
            best_fit = i;
        }
    }
    if (best_out) *best_out = best;
    return best_fit;
}

/* CPUPhotomosaicGenerator::generateBestFits, CPUPhotomosaicGenerator.cpp:33-113,
 * one size step: strictly sequential raster order over the padded grid
 * (:63,:70); each valid cell sees the already-chosen neighbours through
 * calculateRepeats. */
int orc_generate_best_fits(const orc_shape *s, const float *img, int rows, int cols,
                           const float *lib, int64_t n_lib, const uint8_t *masks,
                           int type, int repeat_range, int repeat_addition,
                           int grid_w, int grid_h, int64_t *grid, double *scores,
                           int max_cells, int early_exit)
{
    const int size = s->size;
    const size_t px = (size_t)size * size;
    float *cell = (float *)malloc(sizeof(float) * 3u * px);
    const int rep_cap = repeat_range * (2 * repeat_range + 1) + repeat_range + 1;
    int64_t *rep_keys = (int64_t *)malloc(sizeof(int64_t) * (size_t)(rep_cap > 0 ? rep_cap : 1));
    int *rep_vals = (int *)malloc(sizeof(int) * (size_t)(rep_cap > 0 ? rep_cap : 1));
    int done = 0;

    if (scores)
        for (size_t i = 0; i < (size_t)grid_w * grid_h; ++i) scores[i] = NAN;

    for (int y = -ORC_PAD_GRID; y < grid_h - ORC_PAD_GRID; ++y) {
        for (int x = -ORC_PAD_GRID; x < grid_w - ORC_PAD_GRID; ++x) {
            const size_t pos = (size_t)(y + ORC_PAD_GRID) * grid_w + (x + ORC_PAD_GRID);
            if (grid[pos] < 0)
                continue;
            if (max_cells >= 0 && done >= max_cells)
                goto out;

            const orc_rect bounds = orc_cell_at(s, x, y, img, rows, cols, cell);
            int fh, fv;
            orc_flip_state_at(s, x, y, &fh, &fv);
            const uint8_t *mask = masks + (size_t)(fh + 2 * fv) * px;
            const int n_rep = orc_repeats(grid, grid_w, grid_h, x + ORC_PAD_GRID,
                                          y + ORC_PAD_GRID, repeat_range, repeat_addition,
                                          rep_keys, rep_vals, rep_cap);
            double best;
            grid[pos] = orc_find_cell_best_fit(type, cell, bounds, mask, size, lib, n_lib,
                                               rep_keys, rep_vals, n_rep, early_exit, &best);
            if (scores) scores[pos] = best;
            ++done;
        }
    }
out:
    free(cell);
    free(rep_keys);
    free(rep_vals);
    return done;
}

/* PhotomosaicGeneratorBase::buildPhotomosaic, PhotomosaicGeneratorBase.cpp:110-207,
 * single size step (so `mosaic = mosaicStep`, :197-201).  Library images get an
 * alpha channel of 255 (ImageUtility::addAlphaChannel, Other/ImageUtility.cpp:173-186). */
void orc_build_photomosaic(const orc_shape *s, int rows, int cols, const uint8_t *lib_bgr,
                           const uint8_t *masks, int grid_w, int grid_h, const int64_t *grid,
                           const uint8_t background[4], uint8_t *mosaic_out)
{
    const int size = s->size;
    const size_t px = (size_t)size * size;
    for (size_t i = 0; i < (size_t)rows * cols; ++i)
        memcpy(mosaic_out + 4 * i, background, 4);

    for (int y = -ORC_PAD_GRID; y < grid_h - ORC_PAD_GRID; ++y) {
        for (int x = -ORC_PAD_GRID; x < grid_w - ORC_PAD_GRID; ++x) {
            const int64_t fit = grid[(size_t)(y + ORC_PAD_GRID) * grid_w + (x + ORC_PAD_GRID)];
            if (fit < 0)
                continue;                                          /* :147-150 */
            const orc_rect g = orc_rect_at(s, x, y);
            const int y0 = clampi(g.y, 0, rows), y1 = clampi(g.y + g.h, 0, rows);
            const int x0 = clampi(g.x, 0, cols), x1 = clampi(g.x + g.w, 0, cols);
            int fh, fv;
            orc_flip_state_at(s, x, y, &fh, &fv);
            const uint8_t *mask = masks + (size_t)(fh + 2 * fv) * px;
            const uint8_t *im = lib_bgr + (size_t)fit * px * 3u;
            for (int r = y0; r < y1; ++r) {
                for (int c = x0; c < x1; ++c) {
                    const size_t local = (size_t)(r - g.y) * size + (c - g.x);
                    if (mask[local] == 0)
                        continue;
                    uint8_t *dst = mosaic_out + ((size_t)r * cols + c) * 4u;
                    dst[0] = im[3 * local];
                    dst[1] = im[3 * local + 1];
                    dst[2] = im[3 * local + 2];
                    dst[3] = 255;
                }
            }
        }
    }
}
