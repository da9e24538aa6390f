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

#include <float.h>
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
/* cv::resize INTER_AREA, cv::cvtColor BGR2GRAY (OpenCV 4.x imgproc -- absent   */
/* from the reference checkout; restated from the library's published source:  */
/* resize.cpp ResizeAreaFastVec / ResizeAreaFast_ / computeResizeAreaTab /      */
/* ResizeArea_Invoker, color_yuv RGB2Gray<uchar>).  PARITY UNPINNED at these    */
/* steps: no reference test holds a vector for them.                           */
/* ------------------------------------------------------------------------ */

/* INTER_AREA by an integer factor f on f32 HWC C3: box mean, float row sums left to
 * right, rows top to bottom, times 1/(f f)  (f = 2: ((a + b) + (c + d)) * 0.25f). */
void orc_shrink_f32(const float *src, int rows, int cols, int f, float *dst)
{
    const int ro = rows / f, co = cols / f;
    const float inv = 1.0f / (float)(f * f);
    for (int r = 0; r < ro; ++r)
        for (int c = 0; c < co; ++c)
            for (int ch = 0; ch < 3; ++ch) {
                float total = 0.f;
                for (int i = 0; i < f; ++i) {
                    float row = 0.f;
                    for (int j = 0; j < f; ++j)
                        row += src[((size_t)(r * f + i) * cols + (c * f + j)) * 3u + ch];
                    total += row;
                }
                dst[((size_t)r * co + c) * 3u + ch] = total * inv;
            }
}

/* INTER_AREA by an integer factor f on u8 HWC with cn channels: f = 2: (sum + 2) >> 2
 * (ResizeAreaFastVec<uchar>), otherwise cvRound(float(sum) * (1.f / (f f))). */
void orc_shrink_u8(const uint8_t *src, int rows, int cols, int cn, int f, uint8_t *dst)
{
    const int ro = rows / f, co = cols / f;
    const float scale = 1.0f / (float)(f * f);
    for (int r = 0; r < ro; ++r)
        for (int c = 0; c < co; ++c)
            for (int ch = 0; ch < cn; ++ch) {
                int sum = 0;
                for (int i = 0; i < f; ++i)
                    for (int j = 0; j < f; ++j)
                        sum += src[((size_t)(r * f + i) * cols + (c * f + j)) * cn + ch];
                int v = f == 2 ? (sum + 2) >> 2 : (int)lrintf((float)sum * scale);
                dst[((size_t)r * co + c) * cn + ch] = (uint8_t)clampi(v, 0, 255);
            }
}

/* One hue-rotated colour-scheme variant (ColourScheme.cpp:44-177): u8 BGR -> f32 ->
 * BGR2HSV_FULL -> H = fmod(H + degrees, 360) -> HSV2BGR_FULL -> u8, in f32, following
 * OpenCV 4.5.2's documented float formulas (RGB2HSV_f / HSV2RGB_native); OpenCV is absent
 * from the checkout and no reference test holds vectors: parity unpinned at this step. */
void orc_hue_rotate_u8(const uint8_t *src, size_t n_pixels, float degrees, uint8_t *dst)
{
    static const int sector_data[6][3] = {{1, 3, 0}, {1, 0, 2}, {3, 0, 1}, {0, 2, 1}, {0, 1, 3}, {2, 1, 0}};
    for (size_t t = 0; t < n_pixels; ++t) {
        const float b = (float)src[3 * t], g = (float)src[3 * t + 1], r = (float)src[3 * t + 2];
        float v = r, vmin = r, diff, s, h;
        if (v < g) v = g;
        if (v < b) v = b;
        if (vmin > g) vmin = g;
        if (vmin > b) vmin = b;
        diff = v - vmin;
        s = diff / (fabsf(v) + FLT_EPSILON);
        diff = 60.0f / (diff + FLT_EPSILON);
        if (v == r) h = (g - b) * diff;
        else if (v == g) h = (b - r) * diff + 120.0f;
        else h = (r - g) * diff + 240.0f;
        if (h < 0.0f) h += 360.0f;
        h = fmodf(h + degrees, 360.0f);              /* ColourScheme.cpp:52 */
        float o[3];
        if (s == 0.0f) {
            o[0] = o[1] = o[2] = v;
        } else {
            float hh = fmodf(h * (6.0f / 360.0f), 6.0f);
            int sector = (int)floorf(hh);
            hh -= (float)sector;
            if ((unsigned)sector >= 6u) { sector = 0; hh = 0.0f; }
            const float tab[4] = {v, v * (1.0f - s), v * (1.0f - s * hh), v * (1.0f - s * (1.0f - hh))};
            o[0] = tab[sector_data[sector][0]]; o[1] = tab[sector_data[sector][1]]; o[2] = tab[sector_data[sector][2]];
        }
        for (int k = 0; k < 3; ++k) dst[3 * t + k] = (uint8_t)clampi((int)lrintf(o[k]), 0, 255);
    }
}

typedef struct { int si, di; float alpha; } area_tab;

/* computeResizeAreaTab: source cells (and their weights) under each destination cell. */
static int area_table(int ssize, int dsize, double scale, area_tab *tab)
{
    int k = 0;
    for (int dx = 0; dx < dsize; ++dx) {
        const double fsx1 = dx * scale, fsx2 = fsx1 + scale;
        const double cell = scale < ssize - fsx1 ? scale : ssize - fsx1;
        int sx1 = (int)ceil(fsx1), sx2 = (int)floor(fsx2);
        if (sx2 > ssize - 1) sx2 = ssize - 1;
        if (sx1 > sx2) sx1 = sx2;
        if (sx1 - fsx1 > 1e-3) { tab[k].di = dx; tab[k].si = sx1 - 1; tab[k++].alpha = (float)((sx1 - fsx1) / cell); }
        for (int sx = sx1; sx < sx2; ++sx) { tab[k].di = dx; tab[k].si = sx; tab[k++].alpha = (float)(1.0 / cell); }
        if (fsx2 - sx2 > 1e-3) {
            double w = fsx2 - sx2; if (w > 1.0) w = 1.0; if (w > cell) w = cell;
            tab[k].di = dx; tab[k].si = sx2; tab[k++].alpha = (float)(w / cell);
        }
    }
    return k;
}

/* cv::resize(src, dst, Size(dc, dr), 0, 0, INTER_AREA) for a SHRINK (dr <= sr, dc <= sc)
 * of u8 HWC with cn channels: the integer-factor fast path when both scales are whole,
 * else the weighted form (float accumulators, horizontal pass then vertical, cvRound). */
void orc_resize_area_u8(const uint8_t *src, int sr, int sc, int cn, int dr, int dc, uint8_t *dst)
{
    if (sr == dr && sc == dc) { memcpy(dst, src, (size_t)sr * sc * cn); return; }
    if (sr % dr == 0 && sc % dc == 0 && sr / dr == sc / dc) { orc_shrink_u8(src, sr, sc, cn, sr / dr, dst); return; }
    const double scale_x = 1.0 / ((double)dc / sc), scale_y = 1.0 / ((double)dr / sr);
    area_tab *xt = (area_tab *)malloc(sizeof(area_tab) * 2u * (size_t)sc);
    area_tab *yt = (area_tab *)malloc(sizeof(area_tab) * 2u * (size_t)sr);
    const int nx = area_table(sc, dc, scale_x, xt), ny = area_table(sr, dr, scale_y, yt);
    float *buf = (float *)malloc(sizeof(float) * (size_t)dc * cn);
    float *sum = (float *)calloc((size_t)dc * cn, sizeof(float));
    int prev = yt[0].di;
    for (int j = 0; j < ny; ++j) {
        const float beta = yt[j].alpha;
        const uint8_t *row = src + (size_t)yt[j].si * sc * cn;
        for (int i = 0; i < dc * cn; ++i) buf[i] = 0.f;
        for (int k = 0; k < nx; ++k)
            for (int ch = 0; ch < cn; ++ch)
                buf[xt[k].di * cn + ch] += (float)row[xt[k].si * cn + ch] * xt[k].alpha;
        if (yt[j].di != prev) {
            for (int i = 0; i < dc * cn; ++i) {
                dst[(size_t)prev * dc * cn + i] = (uint8_t)clampi((int)lrintf(sum[i]), 0, 255);
                sum[i] = beta * buf[i];
            }
            prev = yt[j].di;
        } else {
            for (int i = 0; i < dc * cn; ++i) sum[i] += beta * buf[i];
        }
    }
    for (int i = 0; i < dc * cn; ++i) dst[(size_t)prev * dc * cn + i] = (uint8_t)clampi((int)lrintf(sum[i]), 0, 255);
    free(xt); free(yt); free(buf); free(sum);
}

/* cv::cvtColor(BGR2GRAY) on u8: (B 1868 + G 9617 + R 4899 + 2^13) >> 14. */
static uint8_t gray_of(const uint8_t *bgr)
{
    return (uint8_t)((bgr[0] * 1868 + bgr[1] * 9617 + bgr[2] * 4899 + (1 << 13)) >> 14);
}

/* ImageUtility::calculateEntropy, Other/ImageUtility.cpp:189-242: histogram of the grey
 * values under the mask (mask may be NULL), entropy = -sum p log2 p. */
double orc_entropy(const uint8_t *bgr, int rows, int cols, const uint8_t *mask, int mask_stride)
{
    size_t hist[256] = {0}, count = 0;
    for (int r = 0; r < rows; ++r)
        for (int c = 0; c < cols; ++c)
            if (!mask || mask[(size_t)r * mask_stride + c] != 0) {
                ++hist[gray_of(bgr + ((size_t)r * cols + c) * 3u)];
                ++count;
            }
    double entropy = 0;
    for (int i = 0; i < 256; ++i) {
        const double p = hist[i] / (double)count;
        if (p > 0) entropy -= p * log2(p);
    }
    return entropy;
}

/* cv::resize(src, dst, Size(dc, dr), 0, 0, INTER_AREA) for a shrink of f32 HWC with cn channels
 * by a ratio that is not whole: computeResizeAreaTab + ResizeArea_Invoker<float, float>
 * (float accumulators, horizontal pass in table order, then down the contributing rows). */
void orc_resize_area_f32(const float *src, int sr, int sc, int cn, int dr, int dc, float *dst)
{
    const double scale_x = 1.0 / ((double)dc / sc), scale_y = 1.0 / ((double)dr / sr);
    area_tab *xt = (area_tab *)malloc(sizeof(area_tab) * 2u * (size_t)sc);
    area_tab *yt = (area_tab *)malloc(sizeof(area_tab) * 2u * (size_t)sr);
    const int nx = area_table(sc, dc, scale_x, xt), ny = area_table(sr, dr, scale_y, yt);
    float *buf = (float *)malloc(sizeof(float) * (size_t)dc * cn);
    float *sum = (float *)calloc((size_t)dc * cn, sizeof(float));
    int prev = yt[0].di;
    for (int j = 0; j < ny; ++j) {
        const float beta = yt[j].alpha;
        const float *row = src + (size_t)yt[j].si * sc * cn;
        for (int i = 0; i < dc * cn; ++i) buf[i] = 0.f;
        for (int k = 0; k < nx; ++k)
            for (int ch = 0; ch < cn; ++ch)
                buf[xt[k].di * cn + ch] += row[xt[k].si * cn + ch] * xt[k].alpha;
        if (yt[j].di != prev) {
            for (int i = 0; i < dc * cn; ++i) {
                dst[(size_t)prev * dc * cn + i] = sum[i];
                sum[i] = beta * buf[i];
            }
            prev = yt[j].di;
        } else {
            for (int i = 0; i < dc * cn; ++i) sum[i] += beta * buf[i];
        }
    }
    for (int i = 0; i < dc * cn; ++i) dst[(size_t)prev * dc * cn + i] = sum[i];
    free(xt); free(yt); free(buf); free(sum);
}

/* PhotomosaicGeneratorBase::getCellAt, PhotomosaicGeneratorBase.cpp:293-329, at any
 * detail (a whole cell-size ratio is the box mean, any other the general INTER_AREA): the visible part goes into a zero
 * cell of the NORMAL size (:305-310), which is resized to the detail size (:313-315,
 * INTER_AREA), and the local bound is scaled by `detail` with int truncation and the
 * min / max guards (:318-326). */
orc_rect orc_cell_at_detail(const orc_shape *s, int detail_size, double detail, int x, int y,
                            const float *img, int rows, int cols, float *cell_out)
{
    const orc_rect g = orc_rect_at(s, x, y);
    const int size = s->size;
    const int y0 = clampi(g.y, 0, rows), y1 = clampi(g.y + g.h, 0, rows);
    const int x0 = clampi(g.x, 0, cols), x1 = clampi(g.x + g.w, 0, cols);
    const orc_rect local = { x0 - g.x, y0 - g.y, x1 - x0, y1 - y0 };

    float *full = detail_size == size ? cell_out : (float *)malloc(sizeof(float) * 3u * (size_t)size * size);
    memset(full, 0, sizeof(float) * 3u * (size_t)size * (size_t)size);
    for (int r = 0; r < local.h; ++r) {
        const float *src = img + ((size_t)(y0 + r) * cols + x0) * 3u;
        float *dst = full + ((size_t)(local.y + r) * size + local.x) * 3u;
        memcpy(dst, src, sizeof(float) * 3u * (size_t)local.w);
    }
    if (full != cell_out) {
        if (size % detail_size == 0) orc_shrink_f32(full, size, size, size / detail_size, cell_out);
        else orc_resize_area_f32(full, size, size, 3, detail_size, detail_size, cell_out);
        free(full);
    }

    const int ds = detail_size;
    orc_rect b;
    b.x = (int)(local.x * detail); if (b.x > ds - 1) b.x = ds - 1;
    b.y = (int)(local.y * detail); if (b.y > ds - 1) b.y = ds - 1;
    b.w = (int)(local.w * detail); if (b.w < 1) b.w = 1;
    b.h = (int)(local.h * detail); if (b.h < 1) b.h = 1;
    if (b.w > ds - b.x) b.w = ds - b.x;
    if (b.h > ds - b.y) b.h = ds - b.y;
    return b;
}

/* GridBounds::mergeBounds, Grid/GridBounds.cpp:36-107.  Returns the new count. */
static orc_rect rect_union(orc_rect a, orc_rect b)
{
    const int x0 = a.x < b.x ? a.x : b.x, y0 = a.y < b.y ? a.y : b.y;
    const int x1 = a.x + a.w > b.x + b.w ? a.x + a.w : b.x + b.w;
    const int y1 = a.y + a.h > b.y + b.h ? a.y + a.h : b.y + b.h;
    const orc_rect u = { x0, y0, x1 - x0, y1 - y0 };
    return u;
}
static int rect_eq(orc_rect a, orc_rect b) { return a.x == b.x && a.y == b.y && a.w == b.w && a.h == b.h; }

int orc_merge_bounds(orc_rect *b, int n)
{
    int merged = 1;
    while (merged) {
        merged = 0;
        for (int i = 0; i + 1 < n;) {                      /* firstIt + 1 != end */
            for (int j = i + 1; j < n && i + 1 < n;) {
                int merge = 0;
                if (b[i].x == b[j].x && b[i].w == b[j].w) {
                    const int d = b[j].y - b[i].y;
                    if (d == 0) merge = 1;
                    else if (d > 0 && d <= b[i].h) merge = 1;
                    else if (d < 0 && -d <= b[j].h) merge = 1;
                } else if (b[i].y == b[j].y && b[i].h == b[j].h) {
                    const int d = b[j].x - b[i].x;
                    if (d == 0) merge = 1;
                    else if (d > 0 && d <= b[i].w) merge = 1;
                    else if (d < 0 && -d <= b[j].w) merge = 1;
                } else {
                    const orc_rect u = rect_union(b[i], b[j]);
                    if (rect_eq(u, b[i]) || rect_eq(u, b[j])) merge = 1;
                }
                if (merge) {
                    b[i] = rect_union(b[i], b[j]);
                    memmove(b + j, b + j + 1, sizeof(orc_rect) * (size_t)(n - j - 1));
                    --n;
                    merged = 1;
                } else {
                    ++j;
                }
            }
            if (i + 1 < n) ++i; else break;
        }
    }
    return n;
}

/* GridGenerator::findCellState, Grid/GridGenerator.cpp:113-193.  Returns the state
 * (0 = optional(0), -1 = nullopt) and sets *split when the entropy threshold is met. */
static int64_t cell_state(const orc_shape *s, int detail_size, double detail, const uint8_t *masks,
                          const uint8_t *img, int rows, int cols, int x, int y,
                          const orc_rect *bounds, int n_bounds, int not_last_step, int *split)
{
    const orc_rect g = orc_rect_at(s, x, y);
    *split = 0;
    int in_bounds = 0;
    for (int i = 0; i < n_bounds && !in_bounds; ++i) {
        const orc_rect *b = &bounds[i];
        const int y0 = clampi(g.y, b->y, b->y + b->h), y1 = clampi(g.y + g.h, b->y, b->y + b->h);
        const int x0 = clampi(g.x, b->x, b->x + b->w), x1 = clampi(g.x + g.w, b->x, b->x + b->w);
        if (y0 != y1 && x0 != x1) in_bounds = 1;
    }
    if (!in_bounds) return -1;

    if (img && not_last_step) {
        const int y0 = clampi(g.y, 0, rows), y1 = clampi(g.y + g.h, 0, rows);
        const int x0 = clampi(g.x, 0, cols), x1 = clampi(g.x + g.w, 0, cols);
        const orc_rect local = { x0 - g.x, y0 - g.y, x1 - x0, y1 - y0 };
        int fh, fv;
        orc_flip_state_at(s, x, y, &fh, &fv);
        const int ds = detail_size;
        orc_rect b;
        b.x = (int)(local.x * detail); if (b.x > ds - 1) b.x = ds - 1;
        b.y = (int)(local.y * detail); if (b.y > ds - 1) b.y = ds - 1;
        b.w = (int)(local.w * detail); if (b.w < 1) b.w = 1;
        b.h = (int)(local.h * detail); if (b.h < 1) b.h = 1;
        if (b.w > ds - b.x) b.w = ds - b.x;
        if (b.h > ds - b.y) b.h = ds - b.y;
        /* visible part of the image, resized to the bounded mask's size (:153,:180-181) */
        const int vr = y1 - y0, vc = x1 - x0;
        uint8_t *vis = (uint8_t *)malloc((size_t)vr * vc * 3u);
        for (int r = 0; r < vr; ++r)
            memcpy(vis + (size_t)r * vc * 3u, img + ((size_t)(y0 + r) * cols + x0) * 3u, (size_t)vc * 3u);
        uint8_t *cell = vis;
        if (vr != b.h || vc != b.w) {
            /* resizeImage EXACT (Other/ImageUtility.cpp:34-61): INTER_AREA when the factor
             * (by height, or by width when the height matches) is < 1; a visible part
             * SMALLER than the mask (INTER_CUBIC) cannot happen for detail <= 1. */
            cell = (uint8_t *)malloc((size_t)b.h * b.w * 3u);
            orc_resize_area_u8(vis, vr, vc, 3, b.h, b.w, cell);
        }
        const uint8_t *mask = masks + (size_t)(fh + 2 * fv) * ds * ds + (size_t)b.y * ds + b.x;
        const double e = orc_entropy(cell, b.h, b.w, mask, ds);
        if (cell != vis) free(cell);
        free(vis);
        if (e >= 8.0 * 0.7) { *split = 1; return -1; }      /* MAX_ENTROPY * 0.7, ImageUtility.h:69 */
    }
    return 0;
}

/* GridGenerator::getGridState, Grid/GridGenerator.cpp:29-110, all size steps.
 * shapes[step] / detail_sizes[step] / masks[step] (4 detail masks each); img u8 BGR or
 * NULL (no image: every cell valid at step 0, no entropy).  states[step] must hold
 * grid_w[step] * grid_h[step] entries (orc_grid_size).  Returns the number of steps
 * produced (the loop stops when a step leaves no bounds). */
int orc_grid_state_steps(const orc_shape *shapes, const int *detail_sizes, const uint8_t *const *masks,
                         int size_steps, double detail, const uint8_t *img, int rows, int cols,
                         int64_t *const *states, const int *grid_w, const int *grid_h)
{
    int cap = 16, n_active = 1, n_next = 0, produced = 0;
    orc_rect *active = (orc_rect *)malloc(sizeof(orc_rect) * (size_t)cap);
    orc_rect *next = (orc_rect *)malloc(sizeof(orc_rect) * (size_t)cap);
    active[0].x = 0; active[0].y = 0; active[0].w = cols; active[0].h = rows;
    for (int step = 0; step <= size_steps && n_active > 0; ++step) {
        const orc_shape *s = &shapes[step];
        n_next = 0;
        for (int y = -ORC_PAD_GRID; y < grid_h[step] - ORC_PAD_GRID; ++y)
            for (int x = -ORC_PAD_GRID; x < grid_w[step] - ORC_PAD_GRID; ++x) {
                int split;
                states[step][(size_t)(y + ORC_PAD_GRID) * grid_w[step] + (x + ORC_PAD_GRID)] =
                    cell_state(s, detail_sizes[step], detail, masks[step], img, rows, cols, x, y,
                               active, n_active, step < size_steps, &split);
                if (!split) continue;
                const orc_rect g = orc_rect_at(s, x, y);
                const int y0 = clampi(g.y, 0, rows), y1 = clampi(g.y + g.h, 0, rows);
                const int x0 = clampi(g.x, 0, cols), x1 = clampi(g.x + g.w, 0, cols);
                if (y0 == y1 || x0 == x1) continue;
                if (n_next == cap) {
                    cap *= 2;
                    active = (orc_rect *)realloc(active, sizeof(orc_rect) * (size_t)cap);
                    next = (orc_rect *)realloc(next, sizeof(orc_rect) * (size_t)cap);
                }
                next[n_next].x = x0; next[n_next].y = y0; next[n_next].w = x1 - x0; next[n_next].h = y1 - y0;
                ++n_next;
            }
        ++produced;
        orc_rect *t = active; active = next; next = t;
        n_active = n_next;
        if (n_active > 0) n_active = orc_merge_bounds(active, n_active);
    }
    free(active); free(next);
    return produced;
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
int orc_generate_step(const orc_shape *s, int detail_size, double detail,
                      const float *img, int rows, int cols,
                      const float *lib, int64_t n_lib, const uint8_t *masks,
                      int type, int repeat_range, int repeat_addition,
                      int grid_w, int grid_h, int64_t *grid, double *scores,
                      int max_cells, int early_exit)
{
    const int size = detail_size;
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

            const orc_rect bounds = orc_cell_at_detail(s, detail_size, detail, x, y, img, rows, cols, cell);
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

int orc_generate_best_fits(const orc_shape *s, const float *img, int rows, int cols,
                           const float *lib, int64_t n_lib, const uint8_t *masks,
                           int type, int repeat_range, int repeat_addition,
                           int grid_w, int grid_h, int64_t *grid, double *scores,
                           int max_cells, int early_exit)
{
    return orc_generate_step(s, s->size, 1.0, img, rows, cols, lib, n_lib, masks, type,
                             repeat_range, repeat_addition, grid_w, grid_h, grid, scores,
                             max_cells, early_exit);
}

/* The reference's pick for LISTED cells, each taken as the raster loop of
 * generateBestFits (CPUPhotomosaicGenerator.cpp:63-91) reaches it when every earlier cell
 * holds the pick recorded in `grid`: getCellAt, the flip state's mask, calculateRepeats over
 * `grid` (it reads earlier cells only, :185-225), findCellBestFit.  `grid` is not written.
 * If the answer equals grid[pos] for EVERY valid cell of a raster-order prefix, that prefix
 * of `grid` is what the sequential loop produces (induction over the raster order), which
 * lets a checker verify a finished grid with one independent call per cell, in parallel. */
void orc_picks_given_grid(const orc_shape *s, int detail_size, double detail,
                          const float *img, int rows, int cols,
                          const float *lib, int64_t n_lib, const uint8_t *masks,
                          int type, int repeat_range, int repeat_addition,
                          int grid_w, int grid_h, const int64_t *grid,
                          const int32_t *pos_x, const int32_t *pos_y, int n, int early_exit,
                          int64_t *pick_out, double *score_out)
{
    const int size = detail_size;
    const size_t px = (size_t)size * size;
    float *cell = (float *)malloc(sizeof(float) * 3u * px);
    const int rep_cap = repeat_range * (2 * repeat_range + 1) + repeat_range + 1;
    int64_t *rep_keys = (int64_t *)malloc(sizeof(int64_t) * (size_t)(rep_cap > 0 ? rep_cap : 1));
    int *rep_vals = (int *)malloc(sizeof(int) * (size_t)(rep_cap > 0 ? rep_cap : 1));
    for (int k = 0; k < n; ++k) {
        const int x = pos_x[k] - ORC_PAD_GRID, y = pos_y[k] - ORC_PAD_GRID;   /* padded -> grid coordinates */
        const orc_rect bounds = orc_cell_at_detail(s, detail_size, detail, x, y, img, rows, cols, cell);
        int fh, fv;
        orc_flip_state_at(s, x, y, &fh, &fv);
        const uint8_t *mask = masks + (size_t)(fh + 2 * fv) * px;
        const int n_rep = orc_repeats(grid, grid_w, grid_h, pos_x[k], pos_y[k], repeat_range,
                                      repeat_addition, rep_keys, rep_vals, rep_cap);
        double best;
        pick_out[k] = orc_find_cell_best_fit(type, cell, bounds, mask, size, lib, n_lib,
                                             rep_keys, rep_vals, n_rep, early_exit, &best);
        if (score_out) score_out[k] = best;
    }
    free(cell);
    free(rep_keys);
    free(rep_vals);
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

/* PhotomosaicGeneratorBase::buildPhotomosaic over size steps (PhotomosaicGeneratorBase.cpp:
 * 110-207): shapes[step] = the NORMAL cell shapes, masks[step] their 4 masks, grids[step] the
 * best fits; the u8 library (at shapes[0].size) is halved before every later step (:126-127,
 * orc_shrink_u8) and a later step fills only what no earlier step's masks cover (:189-196). */
void orc_build_photomosaic_steps(const orc_shape *shapes, int n_steps, int rows, int cols,
                                 const uint8_t *lib_bgr, int64_t n_lib, const uint8_t *const *masks,
                                 const int *grid_w, const int *grid_h, const int64_t *const *grids,
                                 const uint8_t background[4], uint8_t *mosaic_out)
{
    const size_t n_px = (size_t)rows * cols;
    uint8_t *step_img = (uint8_t *)malloc(n_px * 4u);
    uint8_t *covered = (uint8_t *)calloc(n_px, 1), *step_mask = (uint8_t *)malloc(n_px);
    int size = shapes[0].size;
    uint8_t *lib = (uint8_t *)malloc((size_t)n_lib * size * size * 3u);
    memcpy(lib, lib_bgr, (size_t)n_lib * size * size * 3u);
    for (int step = 0; step < n_steps; ++step) {
        if (step != 0) {
            const int half = size / 2;
            uint8_t *next = (uint8_t *)malloc((size_t)n_lib * half * half * 3u);
            for (int64_t i = 0; i < n_lib; ++i)
                orc_shrink_u8(lib + (size_t)i * size * size * 3u, size, size, 3, 2, next + (size_t)i * half * half * 3u);
            free(lib);
            lib = next;
            size = half;
        }
        const orc_shape *s = &shapes[step];
        const size_t px = (size_t)size * size;
        orc_build_photomosaic(s, rows, cols, lib, masks[step], grid_w[step], grid_h[step], grids[step],
                              background, step_img);
        /* mosaicMaskStep: union of the painted cells' bounded masks (:182-184) */
        memset(step_mask, 0, n_px);
        for (int y = -ORC_PAD_GRID; y < grid_h[step] - ORC_PAD_GRID; ++y)
            for (int x = -ORC_PAD_GRID; x < grid_w[step] - ORC_PAD_GRID; ++x) {
                if (grids[step][(size_t)(y + ORC_PAD_GRID) * grid_w[step] + (x + ORC_PAD_GRID)] < 0) continue;
                const orc_rect g = orc_rect_at(s, x, y);
                const int y0 = clampi(g.y, 0, rows), y1 = clampi(g.y + g.h, 0, rows);
                const int x0 = clampi(g.x, 0, cols), x1 = clampi(g.x + g.w, 0, cols);
                int fh, fv;
                orc_flip_state_at(s, x, y, &fh, &fv);
                const uint8_t *mask = masks[step] + (size_t)(fh + 2 * fv) * px;
                for (int r = y0; r < y1; ++r)
                    for (int c = x0; c < x1; ++c)
                        step_mask[(size_t)r * cols + c] |= mask[(size_t)(r - g.y) * size + (c - g.x)];
            }
        for (size_t i = 0; i < n_px; ++i)
            if (step == 0 || !covered[i]) {
                memcpy(mosaic_out + 4 * i, step_img + 4 * i, 4);
                covered[i] = step_mask[i];
            }
    }
    free(step_img); free(covered); free(step_mask); free(lib);
}
