// kernels.hpp -- device kernels of the best-fit engine (gfx950 / CDNA4, wave64).
//
// Data layout in HBM (DESIGN.md "Layout"):
//   library   f32 planar  [V][L][3][ppad]   ppad = P rounded up to 256, pad = 0; one copy per
//                                           distinct mask variant, pixels permuted so the
//                                           variant's active pixels come first (launch.hpp)
//   masks     u8          [4][ppad]         pad = 0
//   target    f32 HWC     [rows][cols][3]   as handed over by the host
//   cells     f32 planar  [B][3][ppad]      gathered per batch in the cell's variant order, zero outside the image
//   cmask     u8          [B][ppad]         mask variant AND local bound, per cell
//   scores    f64         [B][L]            base score of every (cell, candidate)
//   cand_*    i32/f64     [N][K]            K lowest (score, index) per cell, ascending
//   grid      i64         [gridH][gridW]    chosen index, sentinel L elsewhere
//
// No MFMA anywhere: the per-pair term is a sqrt / transcendental of the pair,
// not a bilinear form, so the work is VALU + a reduction.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/mosaic_hip_tuning.h"   // (includes mosaic_hip.h)
#include "colour_diff.hpp"
#include "colour_diff_fast.hpp"
#include "launch.hpp"

namespace mm {

constexpr int kBlock = 256;          // 4 waves
constexpr int kWave = 64;
constexpr double kInf = __builtin_huge_val();

// ---------------------------------------------------------------------------
// wave / block reductions
// ---------------------------------------------------------------------------

__device__ __forceinline__ double wave_sum(double v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, kWave);
    return v;
}

struct ScoreIdx { double s; int i; };

// Lexicographic (score, index) order: the lowest index wins a tie, which is what
// the reference's ascending scan with strict < does (CPUPhotomosaicGenerator.cpp:165).
__device__ __forceinline__ bool lex_less(double s, int i, double t, int j)
{
    return s < t || (s == t && i < j);
}

__device__ __forceinline__ ScoreIdx wave_min(ScoreIdx v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const double s = __shfl_xor(v.s, off, kWave);
        const int i = __shfl_xor(v.i, off, kWave);
        if (lex_less(s, i, v.s, v.i)) { v.s = s; v.i = i; }
    }
    return v;
}

// wave_min without the LDS crossbar: the partner's (score, index) through DPP row rotations (four steps leave every lane of a 16-lane
// row with the row's minimum), then the four rows' minima through v_readlane.  Same lexicographic order, same result in every lane.
// ds_bpermute costs a wave that has the SIMD to itself (the wavefront resolve) ~100 cycles of LDS latency per step, 18 of them.
template <int CTRL>
__device__ __forceinline__ void lexmin_dpp_step(double& s, int& i)
{
    const unsigned long long b = __builtin_bit_cast(unsigned long long, s);
    const unsigned int lo = (unsigned int)__builtin_amdgcn_mov_dpp((int)(unsigned int)b, CTRL, 0xf, 0xf, false);
    const unsigned int hi = (unsigned int)__builtin_amdgcn_mov_dpp((int)(unsigned int)(b >> 32), CTRL, 0xf, 0xf, false);
    const int oi = __builtin_amdgcn_mov_dpp(i, CTRL, 0xf, 0xf, false);
    const double os = __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
    if (lex_less(os, oi, s, i)) { s = os; i = oi; }
}

__device__ __forceinline__ ScoreIdx wave_min_dpp(ScoreIdx v)
{
    lexmin_dpp_step<0x128>(v.s, v.i);       // row_ror:8
    lexmin_dpp_step<0x124>(v.s, v.i);       // row_ror:4
    lexmin_dpp_step<0x122>(v.s, v.i);       // row_ror:2
    lexmin_dpp_step<0x121>(v.s, v.i);       // row_ror:1
    const unsigned long long b = __builtin_bit_cast(unsigned long long, v.s);
    ScoreIdx r{v.s, v.i};
    bool first = true;
#pragma unroll
    for (int row = 0; row < 4; ++row) {
        const unsigned int lo = (unsigned int)__builtin_amdgcn_readlane((int)(unsigned int)b, 16 * row);
        const unsigned int hi = (unsigned int)__builtin_amdgcn_readlane((int)(unsigned int)(b >> 32), 16 * row);
        const int oi = __builtin_amdgcn_readlane(v.i, 16 * row);
        const double os = __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
        if (first || lex_less(os, oi, r.s, r.i)) { r.s = os; r.i = oi; }
        first = false;
    }
    return r;
}

// Block-wide lexicographic min; every thread gets the result.  `sh` = 4 slots.
__device__ __forceinline__ ScoreIdx block_min(ScoreIdx v, ScoreIdx* sh)
{
    v = wave_min(v);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    __syncthreads();                       // sh may still be read from a previous call
    if (lane == 0) sh[wave] = v;
    __syncthreads();
    ScoreIdx r = sh[0];
#pragma unroll
    for (int w = 1; w < kBlock / kWave; ++w)
        if (lex_less(sh[w].s, sh[w].i, r.s, r.i)) r = sh[w];
    return r;
}

// ---------------------------------------------------------------------------
// staging kernels
// ---------------------------------------------------------------------------

// HWC [n][P][3] -> planar [n][3][ppad] in a mask variant's pixel order: plane position k
// holds original pixel perm[k] (perm[k] < 0: padding, written as 0).
__global__ void __launch_bounds__(kBlock)
hwc_to_planar_kernel(const float* __restrict__ hwc, float* __restrict__ planar,
                     int n_images, int P, int ppad, const int* __restrict__ perm)
{
    const size_t total = (size_t)n_images * ppad;
    for (size_t t = (size_t)blockIdx.x * kBlock + threadIdx.x; t < total;
         t += (size_t)gridDim.x * kBlock) {
        const size_t img = t / ppad;
        const int k = (int)(t - img * ppad);
        const int p = perm[k];
        float v0 = 0.f, v1 = 0.f, v2 = 0.f;
        if (p >= 0) {
            const float* src = hwc + (img * P + p) * 3;
            v0 = src[0]; v1 = src[1]; v2 = src[2];
        }
        float* dst = planar + img * 3 * (size_t)ppad + k;
        dst[0] = v0;
        dst[ppad] = v1;
        dst[2 * (size_t)ppad] = v2;
    }
}

// The f32 pixel of a cell at detail position (r, c).  scale 1: the target pixel under the
// cell.  scale n > 1 (detail level 1/n): the mean of the n x n block of the zero-filled
// full-size cell -- getCellAt copies the visible part of the image into a zero cell and
// resizes it to the detail size with cv::resize INTER_AREA (PhotomosaicGeneratorBase.cpp:
// 305-315), which for an integer factor is the box mean.  Arithmetic SPECIFIED here (OpenCV
// is not part of the reference checkout): float row sums left to right, rows added top to
// bottom, one multiply by 1/(n n); for n = 2 that is ((a + b) + (c + d)) * 0.25f.
// How a detail pixel is made from the normal-size cell.  scale >= 1: whole ratio, box mean (above).
// scale == 0: any ratio -- cv::resize(..., INTER_AREA) for a shrink that is not a whole number is
// computeResizeAreaTab + ResizeArea_Invoker<float, float> (OpenCV 4 imgproc resize.cpp): per axis a
// table of (source index, weight) entries per destination index; a destination row is
//   buf = ((0 + S[sx0] a0) + S[sx1] a1) + ...   for every contributing source row, in table order,
//   sum = b0 * buf0;  sum += b1 * buf1; ...       down the contributing rows,
// in f32 without fused multiply-adds.  Cells are square, so one table serves both axes:
// entries [first[d], first[d + 1]) belong to destination index d (engine.hip build_area_table).
struct CellResize {
    int scale;
    const int* first;       // [detail_size + 1]
    const int* src;         // [entries] source row / column inside the normal-size cell
    const float* alpha;     // [entries]
};

__device__ __forceinline__ void cell_pixel(const float* __restrict__ target, int rows, int cols,
                                           int y, int x, int r, int c, const CellResize& rs, float out[3])
{
    const int n = rs.scale;
    if (n == 1) {
        const float* s = target + ((size_t)(y + r) * cols + (x + c)) * 3;
        out[0] = s[0]; out[1] = s[1]; out[2] = s[2];
        return;
    }
    if (n == 0) {
        float t0 = 0.f, t1 = 0.f, t2 = 0.f;
        const int ky0 = rs.first[r], ky1 = rs.first[r + 1], kx0 = rs.first[c], kx1 = rs.first[c + 1];
        for (int ky = ky0; ky < ky1; ++ky) {
            const int yy = y + rs.src[ky];
            const float beta = rs.alpha[ky];
            float b0 = 0.f, b1 = 0.f, b2 = 0.f;
            for (int kx = kx0; kx < kx1; ++kx) {
                const int xx = x + rs.src[kx];
                const float al = rs.alpha[kx];
                float s0 = 0.f, s1 = 0.f, s2 = 0.f;                  // the cell is zero outside the image
                if (yy >= 0 && yy < rows && xx >= 0 && xx < cols) {
                    const float* s = target + ((size_t)yy * cols + xx) * 3;
                    s0 = s[0]; s1 = s[1]; s2 = s[2];
                }
                b0 = b0 + s0 * al; b1 = b1 + s1 * al; b2 = b2 + s2 * al;
            }
            if (ky == ky0) { t0 = beta * b0; t1 = beta * b1; t2 = beta * b2; }
            else { t0 = t0 + beta * b0; t1 = t1 + beta * b1; t2 = t2 + beta * b2; }
        }
        out[0] = t0; out[1] = t1; out[2] = t2;
        return;
    }
    float t0 = 0.f, t1 = 0.f, t2 = 0.f;
    for (int i = 0; i < n; ++i) {
        const int yy = y + r * n + i;
        float r0 = 0.f, r1 = 0.f, r2 = 0.f;
        for (int j = 0; j < n; ++j) {
            const int xx = x + c * n + j;
            if (yy >= 0 && yy < rows && xx >= 0 && xx < cols) {
                const float* s = target + ((size_t)yy * cols + xx) * 3;
                r0 += s[0]; r1 += s[1]; r2 += s[2];
            }
        }
        t0 += r0; t1 += r1; t2 += r2;
    }
    const float inv = 1.0f / (float)(n * n);
    out[0] = t0 * inv; out[1] = t1 * inv; out[2] = t2 * inv;
}

// Library resize by an integer factor on f32 images, [n][S][S][3] -> [n][S/f][S/f][3]:
// ImageUtility::batchResizeMat(lib) between size steps (CPUPhotomosaicGenerator.cpp:95-99,
// Other/ImageUtility.cpp:91-101: cv::resize INTER_AREA to round(0.5 * size)).  Same box mean
// as cell_pixel.
__global__ void __launch_bounds__(kBlock)
shrink_f32_kernel(const float* __restrict__ src, float* __restrict__ dst, int n_images, int size_in, int f)
{
    const int so = size_in / f;
    const size_t total = (size_t)n_images * so * so;
    for (size_t t = (size_t)blockIdx.x * kBlock + threadIdx.x; t < total; t += (size_t)gridDim.x * kBlock) {
        const size_t img = t / ((size_t)so * so);
        const int p = (int)(t - img * so * so);
        const int r = p / so, c = p - r * so;
        float px[3];
        cell_pixel(src + img * (size_t)size_in * size_in * 3, size_in, size_in, 0, 0, r, c, CellResize{f, nullptr, nullptr, nullptr}, px);
        float* d = dst + t * 3;
        d[0] = px[0]; d[1] = px[1]; d[2] = px[2];
    }
}

// preprocessMainImage / preprocessLibraryImages on the device (PhotomosaicGeneratorBase.cpp:
// 223-252, :255-290): u8 BGR -> [optional INTER_AREA shrink by f ON THE u8 IMAGE, as the
// reference resizes the library before converting, :262-271] -> f32 BGR in [0,255]
// (convertTo(CV_32F)) or f32 Lab (convertTo(CV_32F, 1/255) + cvtColor(BGR2Lab)).
// Arithmetic SPECIFIED here, OpenCV being absent from the reference checkout:
//   u8 shrink: f = 2: (a + b + c + d + 2) >> 2; f > 2: rint(float(sum) * (1.f / (f f)))
//              (OpenCV 4 imgproc resize.cpp, ResizeAreaFastVec / ResizeAreaFast_);
//   Lab: sRGB decode by a 256-entry fp64 table, the BGR2Lab matrix and D65 white of
//        OpenCV's documentation, f(t) = cbrt(t) above 0.008856 else 7.787 t + 16/116, in
//        fp64, rounded to f32 once -- the same formulae as synth.bgr_u8_to_lab_f32.
__global__ void __launch_bounds__(kBlock)
preprocess_u8_kernel(const uint8_t* __restrict__ src, float* __restrict__ dst, size_t n_images,
                     int rows_in, int cols_in, int f, int lab, const double* __restrict__ srgb)
{
    const int ro = rows_in / f, co = cols_in / f;
    const size_t per = (size_t)ro * co, total = n_images * per;
    for (size_t t = (size_t)blockIdx.x * kBlock + threadIdx.x; t < total; t += (size_t)gridDim.x * kBlock) {
        const size_t img = t / per;
        const size_t p = t - img * per;
        const int r = (int)(p / co), c = (int)(p - (size_t)r * co);
        const uint8_t* s = src + img * (size_t)rows_in * cols_in * 3;
        int v[3];
        if (f == 1) {
            const uint8_t* q = s + ((size_t)r * cols_in + c) * 3;
            v[0] = q[0]; v[1] = q[1]; v[2] = q[2];
        } else {
            int sum[3] = {0, 0, 0};
            for (int i = 0; i < f; ++i)
                for (int j = 0; j < f; ++j) {
                    const uint8_t* q = s + ((size_t)(r * f + i) * cols_in + (c * f + j)) * 3;
                    sum[0] += q[0]; sum[1] += q[1]; sum[2] += q[2];
                }
            const float scale = 1.0f / (float)(f * f);
            for (int k = 0; k < 3; ++k)
                v[k] = f == 2 ? (sum[k] + 2) >> 2 : min(255, max(0, __float2int_rn((float)sum[k] * scale)));
        }
        float* d = dst + t * 3;
        if (!lab) {
            d[0] = (float)v[0]; d[1] = (float)v[1]; d[2] = (float)v[2];
            continue;
        }
        const double b = srgb[v[0]], g = srgb[v[1]], rr = srgb[v[2]];
        const double x = (0.412453 * rr + 0.357580 * g + 0.180423 * b) / 0.950456;
        const double y = 0.212671 * rr + 0.715160 * g + 0.072169 * b;
        const double z = (0.019334 * rr + 0.119193 * g + 0.950227 * b) / 1.088754;
        const double fx = x > 0.008856 ? cbrt(x) : 7.787 * x + 16.0 / 116.0;
        const double fy = y > 0.008856 ? cbrt(y) : 7.787 * y + 16.0 / 116.0;
        const double fz = z > 0.008856 ? cbrt(z) : 7.787 * z + 16.0 / 116.0;
        const double l = y > 0.008856 ? 116.0 * fy - 16.0 : 903.3 * y;
        d[0] = (float)l;
        d[1] = (float)(500.0 * (fx - fy));
        d[2] = (float)(200.0 * (fy - fz));
    }
}

// One hue-rotated variant of a u8 BGR image: ColourScheme::getColourScheme* (ColourScheme.cpp:
// 44-177) builds every variant as  u8 -> f32 (values stay 0..255) -> cvtColor(BGR2HSV_FULL) ->
// H = fmod(H + degrees, 360) -> cvtColor(HSV2BGR_FULL) -> convertTo(u8).  The two cvtColor calls
// are OpenCV (4.5.2, absent from the checkout; no reference test holds vectors for them), so the
// arithmetic is SPECIFIED here, after OpenCV's documented f32 formulas, in f32, this order:
//   V = max(B,G,R); m = min(B,G,R); d = V - m; S = d / (|V| + FLT_EPSILON); k = 60 / (d + FLT_EPSILON)
//   H = (G-B) k          if V == R
//       (B-R) k + 120    else if V == G
//       (R-G) k + 240    else;            H += 360 if H < 0
//   back: S == 0 -> B = G = R = V; else h = fmod(H * (6/360), 6), i = floor(h), f = h - i
//   (i outside 0..5 -> i = 0, f = 0), tab = {V, V(1-S), V(1-S f), V(1-S(1-f))},
//   (B,G,R) = tab[{1,3,0},{1,0,2},{3,0,1},{0,2,1},{0,1,3},{2,1,0}][i]
//   u8 = saturate(round-half-even) -- cv::saturate_cast<uchar>(float).
// Built without fp contraction (engine.hip), so the oracle and the numpy restatement
// (colourscheme.py) agree with it bit for bit.  Parity unpinned at this step, as for Lab.
__device__ __forceinline__ void hue_rotate_px(const uint8_t* q, float degrees, uint8_t* out)
{
    const float b = (float)q[0], g = (float)q[1], r = (float)q[2];
    float v = r, vmin = r;
    if (v < g) v = g;
    if (v < b) v = b;
    if (vmin > g) vmin = g;
    if (vmin > b) vmin = b;
    float diff = v - vmin;
    const float s = diff / (fabsf(v) + 1.1920928955078125e-07f);
    diff = 60.0f / (diff + 1.1920928955078125e-07f);
    float h;
    if (v == r) h = (g - b) * diff;
    else if (v == g) h = (b - r) * diff + 120.0f;
    else h = (r - g) * diff + 240.0f;
    if (h < 0.0f) h += 360.0f;
    h = fmodf(h + degrees, 360.0f);                       // ColourScheme.cpp:52,69-70,...
    float ob, og, orr;
    if (s == 0.0f) {
        ob = og = orr = v;
    } else {
        float hh = fmodf(h * (6.0f / 360.0f), 6.0f);
        int sector = (int)floorf(hh);
        hh -= (float)sector;
        if ((unsigned)sector >= 6u) { sector = 0; hh = 0.0f; }
        const float tab[4] = {v, v * (1.0f - s), v * (1.0f - s * hh), v * (1.0f - s * (1.0f - hh))};
        constexpr int kSector[6][3] = {{1, 3, 0}, {1, 0, 2}, {3, 0, 1}, {0, 2, 1}, {0, 1, 3}, {2, 1, 0}};
        ob = tab[kSector[sector][0]]; og = tab[kSector[sector][1]]; orr = tab[kSector[sector][2]];
    }
    out[0] = (uint8_t)min(255, max(0, __float2int_rn(ob)));
    out[1] = (uint8_t)min(255, max(0, __float2int_rn(og)));
    out[2] = (uint8_t)min(255, max(0, __float2int_rn(orr)));
}

__global__ void __launch_bounds__(kBlock)
hue_rotate_u8_kernel(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst, size_t n_pixels, float degrees)
{
    for (size_t t = (size_t)blockIdx.x * kBlock + threadIdx.x; t < n_pixels; t += (size_t)gridDim.x * kBlock)
        hue_rotate_px(src + t * 3, degrees, dst + t * 3);
}

// cv::resize(..., INTER_AREA) of square u8 BGR images by a ratio that is not a whole number
// (the library at a detail level like 75 %, PhotomosaicGeneratorBase.cpp:262-271): the same
// tables and accumulation order as cell_pixel's scale == 0 form, rounded half-to-even and
// saturated at the end (ResizeArea_Invoker<uchar, float>, cv::saturate_cast<uchar>).
__global__ void __launch_bounds__(kBlock)
resize_area_u8_kernel(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst, size_t n_images,
                      int size_in, int size_out, CellResize rs)
{
    const size_t per = (size_t)size_out * size_out, total = n_images * per;
    for (size_t t = (size_t)blockIdx.x * kBlock + threadIdx.x; t < total; t += (size_t)gridDim.x * kBlock) {
        const size_t img = t / per;
        const size_t p = t - img * per;
        const int r = (int)(p / size_out), c = (int)(p - (size_t)r * size_out);
        const uint8_t* s = src + img * (size_t)size_in * size_in * 3;
        float t0 = 0.f, t1 = 0.f, t2 = 0.f;
        const int ky0 = rs.first[r], ky1 = rs.first[r + 1], kx0 = rs.first[c], kx1 = rs.first[c + 1];
        for (int ky = ky0; ky < ky1; ++ky) {
            const uint8_t* row = s + (size_t)rs.src[ky] * size_in * 3;
            const float beta = rs.alpha[ky];
            float b0 = 0.f, b1 = 0.f, b2 = 0.f;
            for (int kx = kx0; kx < kx1; ++kx) {
                const uint8_t* q = row + (size_t)rs.src[kx] * 3;
                const float al = rs.alpha[kx];
                b0 = b0 + (float)q[0] * al; b1 = b1 + (float)q[1] * al; b2 = b2 + (float)q[2] * al;
            }
            if (ky == ky0) { t0 = beta * b0; t1 = beta * b1; t2 = beta * b2; }
            else { t0 = t0 + beta * b0; t1 = t1 + beta * b1; t2 = t2 + beta * b2; }
        }
        dst[t * 3 + 0] = (uint8_t)min(255, max(0, __float2int_rn(t0)));
        dst[t * 3 + 1] = (uint8_t)min(255, max(0, __float2int_rn(t1)));
        dst[t * 3 + 2] = (uint8_t)min(255, max(0, __float2int_rn(t2)));
    }
}

// getCellAt on the device (PhotomosaicGeneratorBase.cpp:293-329; detail 1/scale):
// the visible part of the target under the cell rect, zero elsewhere, as planes;
// and the per-cell active mask = mask variant AND local bound (the CUDA path's
// imageDifferenceEdge target_area test, CUDA/PhotomosaicGenerator.cu:53-72).
__global__ void __launch_bounds__(kBlock)
gather_cells_kernel(const mm_cell* __restrict__ recs, const float* __restrict__ target,
                    int rows, int cols, CellResize scale, const uint8_t* __restrict__ masks,
                    float* __restrict__ cells, uint8_t* __restrict__ cmask,
                    int size, int /*P*/, int ppad, const int* __restrict__ perm, int4 var_of_mask,
                    uint8_t* __restrict__ cell_var, uint8_t* __restrict__ cell_full,
                    float* __restrict__ cells_f0)
{
    const mm_cell rec = recs[blockIdx.y];
    const int vm[4] = {var_of_mask.x, var_of_mask.y, var_of_mask.z, var_of_mask.w};
    const int v = vm[rec.mask_id];
    const int k = blockIdx.x * kBlock + threadIdx.x;     // position in the variant's pixel order
    if (k == 0) {
        cell_var[blockIdx.y] = (uint8_t)v;
        // not clipped: the local bound is the whole cell, so the active set is the mask itself
        cell_full[blockIdx.y] = rec.row_min == 0 && rec.col_min == 0 && rec.row_max == size && rec.col_max == size;
    }
    if (k >= ppad) return;
    const int p = perm[(size_t)v * ppad + k];            // original pixel, < 0 = padding
    float v0 = 0.f, v1 = 0.f, v2 = 0.f;
    uint8_t m = 0;
    if (p >= 0) {
        const int row = p / size, col = p - row * size;
        if (row >= rec.row_min && row < rec.row_max && col >= rec.col_min && col < rec.col_max) {
            float px[3];
            cell_pixel(target, rows, cols, rec.y, rec.x, row, col, scale, px);
            v0 = px[0]; v1 = px[1]; v2 = px[2];
            m = masks[(size_t)rec.mask_id * ppad + p] != 0;
        }
    }
    float* dst = cells + (size_t)blockIdx.y * 3 * ppad + k;
    dst[0] = v0;
    dst[ppad] = v1;
    dst[2 * (size_t)ppad] = v2;
    cmask[(size_t)blockIdx.y * ppad + k] = m;
    // Channel 0 once more for the fp32 filter pass, with a quiet NaN where the pixel is not
    // active: every filter term of such a pixel comes out as max(NaN, 0) = 0, so the filter
    // kernels need no mask (filter.hip).
    if (cells_f0) cells_f0[(size_t)blockIdx.y * ppad + k] = m ? v0 : __builtin_nanf("");
}

// Angle in degrees wrapped into [-180, 180].
__device__ __forceinline__ float wrap180_deg(float v) { return fmaf(-360.0f, rintf(v * (1.0f / 360.0f)), v); }

// gather_cells_kernel and everything the fp32 filter wants of a cell, in ONE launch: one workgroup per
// (cell, 256-position group), one wave per 64-position block of the plane order.  Beside the planes, the
// active mask and the NaN-marked channel 0 it writes
//   * the seven block statistics of the early-exit bound (what filter.hip group_sums_kernel computes for the
//     library: sum of channel 0, max |channel 0 - 50|, sums of channels 1 and 2, max chroma, hue arc centre
//     and half-width; LAB = false: the three sums only, the rest is CIEDE2000's) -- a wave holds the block,
//     so the statistics are wave reductions of values already in registers -- and an eighth: whether the
//     block's mask pixels are all active in this cell (a clipped cell's bound counts only such blocks);
//   * LAB: the chroma plane and the interleaved copy [group][plane][256] with the prepared operands
//     L / 2 - 25, a, b, C / 50 and NaN in L and a at inactive pixels (filter.hip interleave_kernel).
// Replaces four launches per batch (gather, chroma planes, interleave, group sums).  The statistics feed
// a bound that carries its own slack for the fp32 summation order; nothing here decides a result.
// Grid: (ppad / 256, cells) exactly -- every lane of every wave takes part in the reductions.
template <bool LAB>
__global__ void __launch_bounds__(kBlock)
gather_prepare_kernel(const mm_cell* __restrict__ recs, const float* __restrict__ target,
                      int rows, int cols, CellResize scale, const uint8_t* __restrict__ masks,
                      float* __restrict__ cells, uint8_t* __restrict__ cmask,
                      int size, int /*P*/, int ppad, const int* __restrict__ perm, int4 var_of_mask,
                      uint8_t* __restrict__ cell_var, uint8_t* __restrict__ cell_full,
                      float* __restrict__ cells_f0, float* __restrict__ cell_chroma,
                      float* __restrict__ cells_il, float* __restrict__ stats, size_t stats_stride, int n_blocks)
{
    const int cell = blockIdx.y;
    const mm_cell rec = recs[cell];
    const int vm[4] = {var_of_mask.x, var_of_mask.y, var_of_mask.z, var_of_mask.w};
    const int v = vm[rec.mask_id];
    const int k = blockIdx.x * kBlock + threadIdx.x;     // position in the variant's pixel order (< ppad)
    if (k == 0) {
        cell_var[cell] = (uint8_t)v;
        // 1: the active set is the whole mask; 2: clipped -- the early-exit bound then counts only the blocks whose
        // active pixels are ALL inside the cell's bound (statistic 7 below)
        cell_full[cell] = (rec.row_min == 0 && rec.col_min == 0 && rec.row_max == size && rec.col_max == size) ? 1 : 2;
    }
    const int p = perm[(size_t)v * ppad + k];            // original pixel, < 0 = padding
    float v0 = 0.f, v1 = 0.f, v2 = 0.f;
    bool m = false, in_mask = false;
    if (p >= 0) {
        const int row = p / size, col = p - row * size;
        in_mask = masks[(size_t)rec.mask_id * ppad + p] != 0;
        if (row >= rec.row_min && row < rec.row_max && col >= rec.col_min && col < rec.col_max) {
            float px[3];
            cell_pixel(target, rows, cols, rec.y, rec.x, row, col, scale, px);
            v0 = px[0]; v1 = px[1]; v2 = px[2];
            m = in_mask;
        }
    }
    float* dst = cells + (size_t)cell * 3 * ppad + k;
    dst[0] = v0;
    dst[ppad] = v1;
    dst[2 * (size_t)ppad] = v2;
    cmask[(size_t)cell * ppad + k] = m ? 1 : 0;
    const float nan = __builtin_nanf("");
    const float f0 = m ? v0 : nan;
    cells_f0[(size_t)cell * ppad + k] = f0;
    float ch = 0.f;
    if constexpr (LAB) {
        ch = chroma_of(v1, v2);
        cell_chroma[(size_t)cell * ppad + k] = ch;
        if (cells_il) {
            float* il = cells_il + (size_t)cell * 4 * ppad + (size_t)(k >> 8) * 1024 + (k & 255);
            il[0] = fmaf(0.5f, f0, -25.0f);                          // NaN stays NaN
            il[256] = m ? v1 : nan;
            il[512] = v2;
            il[768] = __fmul_rn(0.02f, ch);
        }
    }

    // block statistics: this wave = block 4 * blockIdx.x + wave of the cell
    float s0 = m ? v0 : 0.f, s1 = m ? v1 : 0.f, s2 = m ? v2 : 0.f;
    float dev = m ? fabsf(v0 - 50.0f) : 0.f, cmax = m ? ch : 0.f;
    int anywhere = (LAB && m && !(ch >= 1e-3f)) ? 1 : 0, any = m ? 1 : 0;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        s0 += __shfl_xor(s0, off, kWave); s1 += __shfl_xor(s1, off, kWave); s2 += __shfl_xor(s2, off, kWave);
        if constexpr (LAB) {
            dev = fmaxf(dev, __shfl_xor(dev, off, kWave));
            cmax = fmaxf(cmax, __shfl_xor(cmax, off, kWave));
            anywhere |= __shfl_xor(anywhere, off, kWave);
            any |= __shfl_xor(any, off, kWave);
        }
    }
    const int lane = threadIdx.x & 63;
    float* o = stats + (size_t)cell * n_blocks + 4 * blockIdx.x + (threadIdx.x >> 6);
    // statistic 7: 1 where every mask pixel of the block is active in this cell (the library's block statistics are
    // over the mask's pixels: only then do both sides of the bound speak of the same pixel set), else 0
    const bool whole_block = __builtin_amdgcn_ballot_w64(in_mask && !m) == 0ull;
    if (lane == 0) o[7 * stats_stride] = whole_block ? 1.0f : 0.0f;
    if constexpr (!LAB) {
        if (lane == 0) { o[0] = s0; o[2 * stats_stride] = s1; o[3 * stats_stride] = s2; }
    } else {
        // every active pixel's hue (both ends of its (1 + G) range) against the direction of the block's sum
        const float ref = atan2f(s2, s1) * 57.29577951308232f;
        float dlo = 180.0f, dhi = -180.0f, far = 0.0f;
        if (m) {
            const float e1 = wrap180_deg(atan2f(v2, v1) * 57.29577951308232f - ref);
            const float e2 = wrap180_deg(atan2f(v2, 1.5f * v1) * 57.29577951308232f - ref);
            dlo = fminf(e1, e2); dhi = fmaxf(e1, e2);
            far = fmaxf(fabsf(e1), fabsf(e2));
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            dlo = fminf(dlo, __shfl_xor(dlo, off, kWave)); dhi = fmaxf(dhi, __shfl_xor(dhi, off, kWave));
            far = fmaxf(far, __shfl_xor(far, off, kWave));
        }
        if (lane == 0) {
            o[0] = s0; o[stats_stride] = dev; o[2 * stats_stride] = s1; o[3 * stats_stride] = s2; o[4 * stats_stride] = cmax;
            const bool open = anywhere || !any || !(far <= 170.0f) || !(fmaf(s1, s1, s2 * s2) >= 1e-6f);
            o[5 * stats_stride] = open ? 0.0f : wrap180_deg(ref + 0.5f * (dlo + dhi) - 275.0f);
            o[6 * stats_stride] = open ? 180.0f : 0.5f * (dhi - dlo) + 0.02f;
        }
    }
}

// Largest grid of kBlock-thread workgroups HIP launches whole: gridDim x blockDim stays below 2^32.
constexpr long long kMaxBlocks256 = (1ll << 24) - 1;

// ---------------------------------------------------------------------------
// score kernel (the dominant kernel): base score of every (cell, candidate)
// ---------------------------------------------------------------------------
//
// One wave owns one (cell, candidate) pair at a time and walks the whole cell:
// lane k takes float4 pixel groups k, k+64, ... of each colour plane (1 KiB
// coalesced per wave-instruction), accumulates the masked colour difference in
// fp64 in a register, and the 64 partial sums are combined with xor-shuffles.
// The 4 waves of a workgroup take 4 candidates against the same cell.
//
// Work decode (1-D grid): cells vary fastest inside a group of `cell_group`
// cells, then candidate tiles, then cell groups -- so the workgroups resident
// at one time share a few candidate tiles (library reuse in L2) rather than
// streaming the whole library once per cell.
template <int TYPE>
__global__ void __launch_bounds__(kBlock)
score_kernel(const float* __restrict__ cells, const uint8_t* __restrict__ cmask,
             const float* __restrict__ lib, double* __restrict__ scores,
             int n_cells, int L, int ppad, int cand_per_block, int cell_group, Variants var)
{
    const int n_tiles = (L + cand_per_block - 1) / cand_per_block;
    const float* const lib_base = lib;
    // a grid may not exceed 2^32 threads (n_cells x n_tiles x 256 does for 29 241 cells x 10 000 images): the launch
    // is capped (score_grid) and the workgroups stride over the (cell, tile) items
    for (long long bid = blockIdx.x; bid < (long long)n_cells * n_tiles; bid += gridDim.x) {
    const long long per_group = (long long)cell_group * n_tiles;
    const int group = (int)(bid / per_group);
    const long long rem = bid - (long long)group * per_group;
    const int cells_here = min(cell_group, n_cells - group * cell_group);
    const int tile = (int)(rem / cells_here);
    const int cell = group * cell_group + (int)(rem - (long long)tile * cells_here);
    if (tile >= n_tiles || cell >= n_cells) continue;

    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int v = var.cell_var[cell];
    const int groups = var.groups[v];                 // float4 groups per lane holding active pixels
    lib = lib_base + (size_t)v * var.lib_stride;
    const float4* c0 = reinterpret_cast<const float4*>(cells + (size_t)cell * 3 * ppad);
    const float4* c1 = c0 + ppad / 4;
    const float4* c2 = c1 + ppad / 4;
    const uint32_t* mk = reinterpret_cast<const uint32_t*>(cmask + (size_t)cell * ppad);

    const int l_end = min((tile + 1) * cand_per_block, L);
    for (int l = tile * cand_per_block + wave; l < l_end; l += kBlock / kWave) {
        const float4* b0 = reinterpret_cast<const float4*>(lib + (size_t)l * 3 * ppad);
        const float4* b1 = b0 + ppad / 4;
        const float4* b2 = b1 + ppad / 4;
        double acc = 0.0;
        for (int j = 0; j < groups; ++j) {
            const int q = lane + kWave * j;
            const uint32_t m = mk[q];
            if (m == 0) continue;
            const float4 p0 = c0[q], p1 = c1[q], p2 = c2[q];
            const float4 q0 = b0[q], q1 = b1[q], q2 = b2[q];
            if (m & 0x000000ffu) acc += parallel_term<TYPE>(Px{p0.x, p1.x, p2.x}, Px{q0.x, q1.x, q2.x});
            if (m & 0x0000ff00u) acc += parallel_term<TYPE>(Px{p0.y, p1.y, p2.y}, Px{q0.y, q1.y, q2.y});
            if (m & 0x00ff0000u) acc += parallel_term<TYPE>(Px{p0.z, p1.z, p2.z}, Px{q0.z, q1.z, q2.z});
            if (m & 0xff000000u) acc += parallel_term<TYPE>(Px{p0.w, p1.w, p2.w}, Px{q0.w, q1.w, q2.w});
        }
        acc = wave_sum(acc);
        if (lane == 0) scores[(size_t)cell * L + l] = acc;
    }
    }
}

// ---------------------------------------------------------------------------
// candidate selection: the K lowest (score, index) of each cell's row
// ---------------------------------------------------------------------------
//
// One workgroup per cell, K rounds; round r takes the lexicographic minimum of
// the entries strictly above round r-1's pick, so nothing is mutated and the
// row is re-read from L2 (40-80 KB).  Slots past the row length get index -1.
//
// Overflow mark: the resolve breaks near-ties by re-summing ALL candidates within
// `tie_band` of the minimum in the reference's order.  If the first candidate
// left OUT of the list is within the band of the last one kept, the list cuts
// through such a cluster; the last index is then stored as -(index + 2) and the
// resolve recomputes that cell's whole row instead of trusting the list.
__global__ void __launch_bounds__(kBlock)
select_kernel(const double* __restrict__ scores, int L, int K, double tie_band,
              int32_t* __restrict__ cand_index, double* __restrict__ cand_score,
              int KW, const int* __restrict__ row_map)
{
    __shared__ ScoreIdx sh[kBlock / kWave];
    const double* row = scores + (size_t)blockIdx.x * L;
    // rows of the output are KW wide (K <= KW filled, the rest empty); row_map scatters
    const size_t out = (size_t)(row_map ? row_map[blockIdx.x] : (int)blockIdx.x) * KW;
    for (int t = K + threadIdx.x; t < KW; t += kBlock) { cand_index[out + t] = -1; cand_score[out + t] = kInf; }
    double prev_s = -kInf;
    int prev_i = -1;
    for (int r = 0; r <= K; ++r) {
        ScoreIdx best{kInf, 0x7fffffff};
        for (int l = threadIdx.x; l < L; l += kBlock) {
            const double s = row[l];
            if (lex_less(prev_s, prev_i, s, l) && lex_less(s, l, best.s, best.i)) { best.s = s; best.i = l; }
        }
        best = block_min(best, sh);
        const bool none = best.i == 0x7fffffff;
        if (r == K) {        // one past the list: does it belong to the last entry's cluster?
            if (!none && threadIdx.x == 0 && best.s <= prev_s + fabs(prev_s) * (4.0 * tie_band))
                cand_index[out + K - 1] = -(prev_i + 2);
            break;
        }
        if (threadIdx.x == 0) {
            cand_index[out + r] = none ? -1 : best.i;
            cand_score[out + r] = none ? kInf : best.s;
        }
        if (none) {          // row exhausted (K > L, or NaN / +inf tail): fill the rest
            for (int t = r + 1 + threadIdx.x; t < K; t += kBlock) {
                cand_index[out + t] = -1;
                cand_score[out + t] = kInf;
            }
            break;
        }
        prev_s = best.s;
        prev_i = best.i;
    }
}

// ---------------------------------------------------------------------------
// tiny cells: detail cells of at most 64 pixels (the reference's harness runs cell sizes down to 8 at detail
// 20 %, Benchmark/Benchmark_Generator.h:163-171: detail cells of 6, 3 and 1 pixels)
// ---------------------------------------------------------------------------
//
// The general path gives every (cell, candidate) a wave and every cell several workgroups; with a handful of
// pixels per cell that is all overhead.  Here ONE WAVE owns a cell from the image to its candidate list:
//   lane k fetches pixel k of the cell straight from the target (getCellAt on the device, as gather_cells_kernel);
//   the candidates are spread over the lanes (l = lane, lane + 64, ...), each lane summing ITS candidate's
//   differences over the active pixels in fp64, reference form, in plane order -- which for cells of up to 8 x 8 is
//   the reference's own row-major order --, the cell's pixel broadcast per step with v_readlane;
//   K + 1 rounds of a wave-wide lexicographic minimum over the row pick the list and the overflow mark
//   (select_kernel's rule), or the whole row in index order when K >= L (full_list_kernel's).
// The library is read from a position-major copy [variant][position][channel][image] (transpose_tiny_kernel), so the
// 64 lanes of a load are 64 consecutive floats.  No filter tier, no seeds: the work is too small to prune.
__global__ void __launch_bounds__(kBlock)
transpose_tiny_kernel(const float* __restrict__ planar, float* __restrict__ out, int n_var, int L, int ppad, int n_pos)
{
    const size_t total = (size_t)n_var * n_pos * 3 * L;
    for (size_t t = (size_t)blockIdx.x * kBlock + threadIdx.x; t < total; t += (size_t)gridDim.x * kBlock) {
        const int l = (int)(t % L);
        size_t r = t / L;
        const int c = (int)(r % 3); r /= 3;
        const int k = (int)(r % n_pos);
        const int v = (int)(r / n_pos);
        out[t] = planar[(((size_t)v * L + l) * 3 + c) * ppad + k];
    }
}

constexpr int kTinyPositions = 64;

template <int TYPE>
__global__ void __launch_bounds__(kBlock)
tiny_cells_kernel(const mm_cell* __restrict__ recs, int n_cells, const float* __restrict__ target, int rows, int cols,
                  CellResize scale, const uint8_t* __restrict__ masks, int size, int ppad, const int* __restrict__ perm,
                  int4 var_of_mask, const float* __restrict__ lib_tiny, int L, double* __restrict__ scores,
                  int K, double tie_band, int32_t* __restrict__ cand_index, double* __restrict__ cand_score,
                  int KW, const int* __restrict__ row_map)
{
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int cell = blockIdx.x * (kBlock / kWave) + wave;
    if (cell >= n_cells) return;
    const mm_cell rec = recs[cell];
    const int vm[4] = {var_of_mask.x, var_of_mask.y, var_of_mask.z, var_of_mask.w};
    const int v = vm[rec.mask_id];
    // lane k: pixel at plane position k of the cell's mask variant (active pixels first)
    const int p = perm[(size_t)v * ppad + lane];
    float v0 = 0.f, v1 = 0.f, v2 = 0.f;
    bool m = false;
    if (p >= 0) {
        const int row = p / size, col = p - row * size;
        if (row >= rec.row_min && row < rec.row_max && col >= rec.col_min && col < rec.col_max &&
            masks[(size_t)rec.mask_id * ppad + p] != 0) {
            float px[3];
            cell_pixel(target, rows, cols, rec.y, rec.x, row, col, scale, px);
            v0 = px[0]; v1 = px[1]; v2 = px[2];
            m = true;
        }
    }
    const unsigned long long active = __builtin_amdgcn_ballot_w64(m);
    double* row_scores = scores + (size_t)cell * L;
    const float* lt = lib_tiny + (size_t)v * kTinyPositions * 3 * L;
    for (int base = 0; base < L; base += kWave) {
        const int l = base + lane;
        double acc = 0.0;
        for (unsigned long long left = active; left != 0ull; left &= left - 1ull) {
            const int k = __builtin_ctzll(left);                                 // wave-uniform
            const float c0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v0), k));
            const float c1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v1), k));
            const float c2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v2), k));
            if (l < L) {
                const float* q = lt + (size_t)k * 3 * L + l;
                acc += parallel_term<TYPE>(Px{c0, c1, c2}, Px{q[0], q[L], q[2 * (size_t)L]});
            }
        }
        if (l < L) row_scores[l] = acc;
    }
    // the list: rows of the output are KW wide (K <= KW filled, the rest empty); row_map scatters
    const size_t out = (size_t)(row_map ? row_map[cell] : cell) * KW;
    for (int t = K + lane; t < KW; t += kWave) { cand_index[out + t] = -1; cand_score[out + t] = kInf; }
    if (K >= L) {                        // the whole row, in index order (full_list_kernel)
        for (int k = lane; k < K; k += kWave) {
            cand_index[out + k] = k < L ? k : -1;
            cand_score[out + k] = k < L ? row_scores[k] : kInf;
        }
        return;
    }
    // (the row was written by this wave's own lanes: visible to them again after the wave-wide exchanges below
    // only through memory, so make it so)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    double prev_s = -kInf;
    int prev_i = -1;
    for (int r = 0; r <= K; ++r) {
        ScoreIdx best{kInf, 0x7fffffff};
        for (int l = lane; l < L; l += kWave) {
            const double sc = row_scores[l];
            if (lex_less(prev_s, prev_i, sc, l) && lex_less(sc, l, best.s, best.i)) { best.s = sc; best.i = l; }
        }
        best = wave_min(best);
        const bool none = best.i == 0x7fffffff;
        if (r == K) {        // one past the list: does it belong to the last entry's cluster?
            if (!none && lane == 0 && best.s <= prev_s + fabs(prev_s) * (4.0 * tie_band))
                cand_index[out + K - 1] = -(prev_i + 2);
            break;
        }
        if (lane == 0) {
            cand_index[out + r] = none ? -1 : best.i;
            cand_score[out + r] = none ? kInf : best.s;
        }
        if (none) {
            for (int t = r + 1 + lane; t < K; t += kWave) { cand_index[out + t] = -1; cand_score[out + t] = kInf; }
            break;
        }
        prev_s = best.s;
        prev_i = best.i;
    }
}

// K >= L (a negative repeat "penalty" can promote any candidate): the list is the whole
// row, in index order; nothing to select.
__global__ void __launch_bounds__(kBlock)
full_list_kernel(const double* __restrict__ scores, int L, int K,
                 int32_t* __restrict__ cand_index, double* __restrict__ cand_score,
                 int KW, const int* __restrict__ row_map)
{
    const double* row = scores + (size_t)blockIdx.x * L;
    const size_t out = (size_t)(row_map ? row_map[blockIdx.x] : (int)blockIdx.x) * KW;
    for (int k = threadIdx.x; k < KW; k += kBlock) {
        cand_index[out + k] = (k < L && k < K) ? k : -1;
        cand_score[out + k] = (k < L && k < K) ? row[k] : kInf;
    }
}

// ---------------------------------------------------------------------------
// two-tier scoring: contenders from the fp32 filter scores, fp64 re-score, re-rank
// ---------------------------------------------------------------------------

struct FilterBand {
    float px_abs;      // bound on the fp32 form's error per pixel pair
    float rel;         // bound on the relative error of the fp32 accumulation
};

// Contenders of one cell: every candidate whose filter score is within the error
// band of the K-th lowest filter score.  If |filter - true| <= err for all
// candidates, the K lowest TRUE scores are all among {filter <= v_K + 2 err}:
// the K candidates with the lowest filter scores have true scores <= v_K + err, so
// the K-th lowest true score is <= v_K + err, and anything truly that low has a
// filter score <= v_K + 2 err.  One workgroup per cell.
//   cont_index [cell][Kc]   candidate ids, -1 padded (unordered)
//   cont_filter[cell][Kc]   their filter scores (for the canary in rerank_kernel)
//   flags[cell]             bit 0: more than Kc contenders (cell needs the dense fp64 path)
__global__ void __launch_bounds__(kBlock)
contenders_kernel(const float* __restrict__ fscores, const uint8_t* __restrict__ cmask,
                  int L, int ppad, int K, int Kc, FilterBand band,
                  int32_t* __restrict__ cont_index, float* __restrict__ cont_filter,
                  uint8_t* __restrict__ flags, int32_t* __restrict__ sentinel, int sample_every)
{
    __shared__ ScoreIdx sh[kBlock / kWave];
    __shared__ int n_cont;
    __shared__ int n_active;
    const float* row = fscores + (size_t)blockIdx.x * L;
    if (threadIdx.x == 0) { n_cont = 0; n_active = 0; }
    __syncthreads();

    // active pixels of the cell (scales the absolute part of the band)
    int act = 0;
    const uint32_t* mk = reinterpret_cast<const uint32_t*>(cmask + (size_t)blockIdx.x * ppad);
    for (int q = threadIdx.x; q < ppad / 4; q += kBlock) act += __popc(mk[q] & 0x01010101u);
    atomicAdd(&n_active, act);

    // K-th lowest filter score (ties by index), K rounds as in select_kernel
    double prev_s = -kInf;
    int prev_i = -1;
    for (int r = 0; r < K; ++r) {
        ScoreIdx best{kInf, 0x7fffffff};
        for (int l = threadIdx.x; l < L; l += kBlock) {
            const double s = (double)row[l];
            if (lex_less(prev_s, prev_i, s, l) && lex_less(s, l, best.s, best.i)) { best.s = s; best.i = l; }
        }
        best = block_min(best, sh);
        if (best.i == 0x7fffffff) break;       // fewer than K finite scores: keep the last one found
        prev_s = best.s;
        prev_i = best.i;
    }
    __syncthreads();
    const double err = (double)n_active * band.px_abs + fabs(prev_s) * band.rel;
    const double thr = (prev_i < 0) ? kInf : prev_s + 2.0 * err;

    int32_t* ci = cont_index + (size_t)blockIdx.x * Kc;
    float* cf = cont_filter + (size_t)blockIdx.x * Kc;
    for (int l = threadIdx.x; l < L; l += kBlock) {
        const float s = row[l];
        if ((double)s <= thr) {
            const int slot = atomicAdd(&n_cont, 1);
            if (slot < Kc) { ci[slot] = l; cf[slot] = s; }
        }
    }
    __syncthreads();
    const int n = n_cont;
    // Run-time guard of the band (sampled cells): the best candidate the band EXCLUDED (finite filter
    // score above the threshold) rides along as a sentinel; rerank_kernel compares its fp64 score with
    // the K-th contender's -- if the excluded one is truly better, the band was too narrow for this
    // content and the cell is re-done fp64-only (flag bit 2) and counted (mm_timing.band_alarms).
    int sent = -1;
    if (sentinel && sample_every > 0 && blockIdx.x % (unsigned)sample_every == 0 && n < Kc) {
        ScoreIdx best{kInf, 0x7fffffff};
        for (int l = threadIdx.x; l < L; l += kBlock) {
            const double sc = (double)row[l];
            if (sc > thr && sc < kInf && lex_less(sc, l, best.s, best.i)) { best.s = sc; best.i = l; }
        }
        best = block_min(best, sh);
        if (best.i != 0x7fffffff) {
            sent = n;
            if (threadIdx.x == 0) { ci[n] = best.i; cf[n] = (float)best.s; }
        }
    }
    for (int t = n + (sent >= 0 ? 1 : 0) + threadIdx.x; t < Kc; t += kBlock) { ci[t] = -1; cf[t] = 0.0f; }
    if (threadIdx.x == 0) {
        flags[blockIdx.x] = n > Kc ? 1 : 0;
        if (sentinel) sentinel[blockIdx.x] = sent;
    }
}

// fp64 reference-form score of every contender: one wave per (cell, slot), same
// lane / pixel assignment and summation order as score_kernel.
template <int TYPE>
__global__ void __launch_bounds__(kBlock)
rescore_kernel(const float* __restrict__ cells, const uint8_t* __restrict__ cmask,
               const float* __restrict__ lib, const int32_t* __restrict__ cont_index,
               double* __restrict__ cont_score, int n_cells, int Kc, int ppad, Variants var)
{
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const long long w = (long long)blockIdx.x * (kBlock / kWave) + wave;
    const int cell = (int)(w / Kc);
    if (cell >= n_cells) return;
    const int l = cont_index[w];
    if (l < 0) { if (lane == 0) cont_score[w] = kInf; return; }

    const int v = var.cell_var[cell];
    const int groups = var.groups[v];
    lib += (size_t)v * var.lib_stride;
    const float4* c0 = reinterpret_cast<const float4*>(cells + (size_t)cell * 3 * ppad);
    const float4* c1 = c0 + ppad / 4;
    const float4* c2 = c1 + ppad / 4;
    const uint32_t* mk = reinterpret_cast<const uint32_t*>(cmask + (size_t)cell * ppad);
    const float4* b0 = reinterpret_cast<const float4*>(lib + (size_t)l * 3 * ppad);
    const float4* b1 = b0 + ppad / 4;
    const float4* b2 = b1 + ppad / 4;
    double acc = 0.0;
    for (int j = 0; j < groups; ++j) {
        const int q = lane + kWave * j;
        const uint32_t m = mk[q];
        if (m == 0) continue;
        const float4 p0 = c0[q], p1 = c1[q], p2 = c2[q];
        const float4 q0 = b0[q], q1 = b1[q], q2 = b2[q];
        if (m & 0x000000ffu) acc += parallel_term<TYPE>(Px{p0.x, p1.x, p2.x}, Px{q0.x, q1.x, q2.x});
        if (m & 0x0000ff00u) acc += parallel_term<TYPE>(Px{p0.y, p1.y, p2.y}, Px{q0.y, q1.y, q2.y});
        if (m & 0x00ff0000u) acc += parallel_term<TYPE>(Px{p0.z, p1.z, p2.z}, Px{q0.z, q1.z, q2.z});
        if (m & 0xff000000u) acc += parallel_term<TYPE>(Px{p0.w, p1.w, p2.w}, Px{q0.w, q1.w, q2.w});
    }
    acc = wave_sum(acc);
    if (lane == 0) cont_score[w] = acc;
}

// Re-rank: the K lowest (fp64 score, index) among a cell's contenders, in the
// format select_kernel produces (overflow mark included).  Also the canary: if any
// contender's filter score is further from its fp64 score than a quarter of the
// band the filter was trusted with, the cell is flagged for the dense fp64 path
// (bit 1) instead of being believed.  One wave per cell would do; a workgroup keeps
// the reduction helpers shared.
__global__ void __launch_bounds__(kBlock)
rerank_kernel(const int32_t* __restrict__ cont_index, const float* __restrict__ cont_filter,
              const double* __restrict__ cont_score, const uint8_t* __restrict__ cmask,
              int Kc, int K, int ppad, FilterBand band, double tie_band,
              int32_t* __restrict__ cand_index, double* __restrict__ cand_score,
              uint8_t* __restrict__ flags, int KW, const int* __restrict__ row_map,
              const int32_t* __restrict__ sentinel, int L)
{
    __shared__ ScoreIdx sh[kBlock / kWave];
    __shared__ int n_active;
    const int sent = sentinel ? sentinel[blockIdx.x] : -1;      // slot of the band guard's excluded candidate, not ranked
    const size_t out = (size_t)(row_map ? row_map[blockIdx.x] : (int)blockIdx.x) * KW;
    for (int t = K + threadIdx.x; t < KW; t += kBlock) { cand_index[out + t] = -1; cand_score[out + t] = kInf; }
    const int32_t* ci = cont_index + (size_t)blockIdx.x * Kc;
    const float* cf = cont_filter + (size_t)blockIdx.x * Kc;
    const double* cs = cont_score + (size_t)blockIdx.x * Kc;
    if (threadIdx.x == 0) n_active = 0;
    __syncthreads();
    int act = 0;
    const uint32_t* mk = reinterpret_cast<const uint32_t*>(cmask + (size_t)blockIdx.x * ppad);
    for (int q = threadIdx.x; q < ppad / 4; q += kBlock) act += __popc(mk[q] & 0x01010101u);
    atomicAdd(&n_active, act);
    __syncthreads();
    if (n_active == 0) {
        // no active pixel: the cell scores exactly 0 against every image, in the reference's own sum too -- the list is
        // the K lowest indices, exact, so neither an overflow mark nor the fp64-only re-run (see rank_kernel)
        for (int t = threadIdx.x; t < K; t += kBlock) {
            cand_index[out + t] = t < L ? t : -1;
            cand_score[out + t] = t < L ? 0.0 : kInf;
        }
        if (threadIdx.x == 0) flags[blockIdx.x] = 0;
        return;
    }

    int bad = 0;
    for (int t = threadIdx.x; t < Kc; t += kBlock) {
        if (ci[t] < 0) continue;
        const double err = (double)n_active * band.px_abs + fabs(cs[t]) * band.rel;
        if (!(fabs((double)cf[t] - cs[t]) <= 0.25 * err)) bad = 1;     // also catches NaN
    }
    if (__syncthreads_or(bad) && threadIdx.x == 0) flags[blockIdx.x] |= 2;

    double prev_s = -kInf;
    int prev_i = -1;
    for (int r = 0; r <= K; ++r) {
        ScoreIdx best{kInf, 0x7fffffff};
        for (int t = threadIdx.x; t < Kc; t += kBlock) {
            const int l = ci[t];
            if (l < 0 || t == sent) continue;
            const double s = cs[t];
            if (lex_less(prev_s, prev_i, s, l) && lex_less(s, l, best.s, best.i)) { best.s = s; best.i = l; }
        }
        best = block_min(best, sh);
        const bool none = best.i == 0x7fffffff;
        // the band failed here if the candidate it EXCLUDED (the sentinel) beats the K-th contender in fp64, or if it
        // excluded a scored candidate although fewer than K contenders were kept
        if (r < K && sent >= 0 && threadIdx.x == 0 && (none || (r == K - 1 && lex_less(cs[sent], ci[sent], best.s, best.i))))
            flags[blockIdx.x] |= 4;
        if (r == K) {
            if (!none && threadIdx.x == 0 && best.s <= prev_s + fabs(prev_s) * (4.0 * tie_band))
                cand_index[out + K - 1] = -(prev_i + 2);
            break;
        }
        if (threadIdx.x == 0) {
            cand_index[out + r] = none ? -1 : best.i;
            cand_score[out + r] = none ? kInf : best.s;
        }
        if (none) {
            for (int t = r + 1 + threadIdx.x; t < K; t += kBlock) {
                cand_index[out + t] = -1;
                cand_score[out + t] = kInf;
            }
            break;
        }
        prev_s = best.s;
        prev_i = best.i;
    }
}

// ---------------------------------------------------------------------------
// one read-back per generate: the result block
// ---------------------------------------------------------------------------
//
// mm_generate's single-submission path enqueues every kernel of a generate without asking the host anything
// in between; whatever the host used to decide from (cells the filter could not vouch for, overflow marks,
// sweeps that still changed picks, speculative picks left unverified) is counted into this header, which
// comes back in the SAME copy as the best-fit grid and the scores that follow it in memory.  All zero (and
// one sweep without a change) = the result is final; anything else sends the call down the step-by-step
// path, which recomputes what the header says cannot be trusted.
constexpr int kFastSweeps = 32;         // sweeps a result block has counters for (a set enqueued at once; the first set of a call is 8 unless the engine learned otherwise)
struct ResultHeader {
    unsigned int n_flagged;              // cells with a filter flag (overflow, canary, band alarm)
    unsigned int n_band_alarm;           // ... of which band alarms
    unsigned int n_marked;               // full-length lists cut through a near-tie cluster
    unsigned int n_insufficient;         // speculative picks the resolve could not prove (no-repeat resolve; sweeps: sweep_insufficient)
    unsigned int edge_pairs;             // discontinuity pairs logged by the filter launch
    unsigned int sweeps;                 // sweeps enqueued
    unsigned int sweep_changes[kFastSweeps];
    unsigned int launch_void;            // the filter launch's void word (launch.hpp kFilterVoidWord): non-zero = its scores are not trusted
    unsigned int pad;
    unsigned long long work[3];          // filter work counters: groups visited, groups of a full visit, bound-dropped
    unsigned long long exact_plain, dense_count, pad64;
    unsigned long long sweep_exact[kFastSweeps];
    unsigned int sweep_insufficient[kFastSweeps];   // unverified speculative picks per sweep (the last enqueued one counts)
    // the filter launch's counter block (launch.hpp kFilterCounterWords), written by the filter kernels themselves
    unsigned int filter_counters[kFilterCounterWords];
};
static_assert(sizeof(ResultHeader) % 16 == 0, "the grid follows the header, 8-byte aligned");

// contenders_kernel + rescore_kernel + rerank_kernel in ONE launch, one workgroup per cell: the contender list
// lives in LDS, the four waves share the contenders' fp64 re-scores (same lane / pixel assignment and summation
// order as rescore_kernel, so the scores are bit-identical to the three-kernel form), then the K lowest are
// ranked from LDS.  Saves two launches per batch and the (cell, empty slot) waves of the separate re-score.
// A cell with NO active pixel (a sliver of an offset shape whose bound misses the mask) scores exactly 0
// against every image, in the reference's own sum too (CPUPhotomosaicGenerator.cpp:143-161 adds nothing):
// its list is the K lowest indices with score 0 -- exact, so no overflow mark and no fp64-only re-run.
// Dynamic LDS: Kc * 16 bytes (f64 scores | i32 indices | f32 filter scores).
template <int TYPE>
__global__ void __launch_bounds__(kBlock)
rank_kernel(const float* __restrict__ fscores, const float* __restrict__ cells, const uint8_t* __restrict__ cmask,
            const float* __restrict__ lib, int L, int ppad, int K, int Kc, FilterBand band, double tie_band,
            int32_t* __restrict__ cand_index, double* __restrict__ cand_score, uint8_t* __restrict__ flags,
            int KW, const int* __restrict__ row_map, int sample_every, Variants var,
            ResultHeader* __restrict__ hdr, int count_marks)
{
    // hdr (may be null): flagged cells -- and, when the lists are final (count_marks), overflow marks -- are counted into
    // the result block's header, so the single-submission path needs no summary launch
    extern __shared__ double rank_lds[];
    double* cs = rank_lds;                                        // [Kc]
    int32_t* ci = reinterpret_cast<int32_t*>(cs + Kc);            // [Kc]
    float* cf = reinterpret_cast<float*>(ci + Kc);                // [Kc]
    __shared__ ScoreIdx sh[kBlock / kWave];
    __shared__ int n_cont;
    __shared__ int n_active;
    const int cell = blockIdx.x;
    const float* row = fscores + (size_t)cell * L;
    const size_t out = (size_t)(row_map ? row_map[cell] : cell) * KW;
    for (int t = K + threadIdx.x; t < KW; t += kBlock) { cand_index[out + t] = -1; cand_score[out + t] = kInf; }
    if (threadIdx.x == 0) { n_cont = 0; n_active = 0; }
    __syncthreads();
    int act = 0;
    const uint32_t* mk = reinterpret_cast<const uint32_t*>(cmask + (size_t)cell * ppad);
    for (int q = threadIdx.x; q < ppad / 4; q += kBlock) act += __popc(mk[q] & 0x01010101u);
    atomicAdd(&n_active, act);
    __syncthreads();
    if (n_active == 0) {
        for (int t = threadIdx.x; t < K; t += kBlock) {
            cand_index[out + t] = t < L ? t : -1;
            cand_score[out + t] = t < L ? 0.0 : kInf;
        }
        if (threadIdx.x == 0) flags[cell] = 0;
        return;
    }

    // K-th lowest filter score (ties by index)
    double prev_s = -kInf;
    int prev_i = -1;
    for (int r = 0; r < K; ++r) {
        ScoreIdx best{kInf, 0x7fffffff};
        for (int l = threadIdx.x; l < L; l += kBlock) {
            const double s = (double)row[l];
            if (lex_less(prev_s, prev_i, s, l) && lex_less(s, l, best.s, best.i)) { best.s = s; best.i = l; }
        }
        best = block_min(best, sh);
        if (best.i == 0x7fffffff) break;
        prev_s = best.s;
        prev_i = best.i;
    }
    const double err_k = (double)n_active * band.px_abs + fabs(prev_s) * band.rel;
    const double thr = (prev_i < 0) ? kInf : prev_s + 2.0 * err_k;
    for (int l = threadIdx.x; l < L; l += kBlock) {
        const float s = row[l];
        if ((double)s <= thr) {
            const int slot = atomicAdd(&n_cont, 1);
            if (slot < Kc) { ci[slot] = l; cf[slot] = s; }
        }
    }
    __syncthreads();
    const int n = n_cont;
    int flag = n > Kc ? 1 : 0;
    // band guard (sampled cells): the best candidate the band EXCLUDED rides along as a sentinel
    int sent = -1;
    if (sample_every > 0 && (unsigned)cell % (unsigned)sample_every == 0 && n < Kc) {
        ScoreIdx best{kInf, 0x7fffffff};
        for (int l = threadIdx.x; l < L; l += kBlock) {
            const double sc = (double)row[l];
            if (sc > thr && sc < kInf && lex_less(sc, l, best.s, best.i)) { best.s = sc; best.i = l; }
        }
        best = block_min(best, sh);
        if (best.i != 0x7fffffff) {
            sent = n;
            if (threadIdx.x == 0) { ci[n] = best.i; cf[n] = (float)best.s; }
        }
    }
    const int n_slots = min(n, Kc) + (sent >= 0 ? 1 : 0);
    __syncthreads();

    // fp64 reference-form score of every contender: one wave per slot
    {
        const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
        const int v = var.cell_var[cell];
        const int groups = var.groups[v];
        const float* vlib = lib + (size_t)v * var.lib_stride;
        const float4* c0 = reinterpret_cast<const float4*>(cells + (size_t)cell * 3 * ppad);
        const float4* c1 = c0 + ppad / 4;
        const float4* c2 = c1 + ppad / 4;
        for (int t = wave; t < n_slots; t += kBlock / kWave) {
            const int l = ci[t];
            const float4* b0 = reinterpret_cast<const float4*>(vlib + (size_t)l * 3 * ppad);
            const float4* b1 = b0 + ppad / 4;
            const float4* b2 = b1 + ppad / 4;
            double acc = 0.0;
            for (int j = 0; j < groups; ++j) {
                const int q = lane + kWave * j;
                const uint32_t m = mk[q];
                if (m == 0) continue;
                const float4 p0 = c0[q], p1 = c1[q], p2 = c2[q];
                const float4 q0 = b0[q], q1 = b1[q], q2 = b2[q];
                if (m & 0x000000ffu) acc += parallel_term<TYPE>(Px{p0.x, p1.x, p2.x}, Px{q0.x, q1.x, q2.x});
                if (m & 0x0000ff00u) acc += parallel_term<TYPE>(Px{p0.y, p1.y, p2.y}, Px{q0.y, q1.y, q2.y});
                if (m & 0x00ff0000u) acc += parallel_term<TYPE>(Px{p0.z, p1.z, p2.z}, Px{q0.z, q1.z, q2.z});
                if (m & 0xff000000u) acc += parallel_term<TYPE>(Px{p0.w, p1.w, p2.w}, Px{q0.w, q1.w, q2.w});
            }
            acc = wave_sum(acc);
            if (lane == 0) cs[t] = acc;
        }
    }
    __syncthreads();

    // canary: a contender whose filter score is further from its fp64 score than a quarter of the band
    int bad = 0;
    for (int t = threadIdx.x; t < n_slots; t += kBlock) {
        const double err = (double)n_active * band.px_abs + fabs(cs[t]) * band.rel;
        if (!(fabs((double)cf[t] - cs[t]) <= 0.25 * err)) bad = 1;     // also catches NaN
    }
    if (__syncthreads_or(bad)) flag |= 2;

    // the K lowest (fp64 score, index), the overflow mark, the band guard's verdict
    prev_s = -kInf;
    prev_i = -1;
    for (int r = 0; r <= K; ++r) {
        ScoreIdx best{kInf, 0x7fffffff};
        for (int t = threadIdx.x; t < n_slots; t += kBlock) {
            if (t == sent) continue;
            const int l = ci[t];
            const double s = cs[t];
            if (lex_less(prev_s, prev_i, s, l) && lex_less(s, l, best.s, best.i)) { best.s = s; best.i = l; }
        }
        best = block_min(best, sh);
        const bool none = best.i == 0x7fffffff;
        if (r < K && sent >= 0 && (none || (r == K - 1 && lex_less(cs[sent], ci[sent], best.s, best.i)))) flag |= 4;
        if (r == K) {
            if (!none && threadIdx.x == 0 && best.s <= prev_s + fabs(prev_s) * (4.0 * tie_band)) {
                cand_index[out + K - 1] = -(prev_i + 2);
                if (hdr && count_marks) atomicAdd(&hdr->n_marked, 1u);
            }
            break;
        }
        if (threadIdx.x == 0) {
            cand_index[out + r] = none ? -1 : best.i;
            cand_score[out + r] = none ? kInf : best.s;
        }
        if (none) {
            for (int t = r + 1 + threadIdx.x; t < K; t += kBlock) {
                cand_index[out + t] = -1;
                cand_score[out + t] = kInf;
            }
            break;
        }
        prev_s = best.s;
        prev_i = best.i;
    }
    if (threadIdx.x == 0) {
        flags[cell] = (uint8_t)flag;
        if (hdr && flag) { atomicAdd(&hdr->n_flagged, 1u); if (flag & 4) atomicAdd(&hdr->n_band_alarm, 1u); }
    }
}

// Header zeroed (with it the filter kernels' counter block: no memset sits in the stream), grid filled with the
// sentinel (CUDAPhotomosaicGenerator.cpp:140-149), scores with NaN; optional byte arrays cleared (list flags).
__global__ void __launch_bounds__(kBlock)
result_init_kernel(ResultHeader* __restrict__ hdr, long long* __restrict__ grid, double* __restrict__ best,
                   size_t cells_total, long long sentinel, uint8_t* __restrict__ clear_a, uint8_t* __restrict__ clear_b, int clear_n)
{
    if (blockIdx.x == 0) {
        unsigned int* w = reinterpret_cast<unsigned int*>(hdr);
        for (int t = threadIdx.x; t < (int)(sizeof(ResultHeader) / 4); t += kBlock) w[t] = 0u;
    }
    for (int t = blockIdx.x * kBlock + threadIdx.x; t < clear_n; t += gridDim.x * kBlock) {
        if (clear_a) clear_a[t] = 0;
        if (clear_b) clear_b[t] = 0;
    }
    const double nan = __builtin_nan("");
    for (size_t t = (size_t)blockIdx.x * kBlock + threadIdx.x; t < cells_total; t += (size_t)gridDim.x * kBlock) {
        grid[t] = sentinel;
        best[t] = nan;
    }
}

// Same fill for the step-by-step resolve (no header).
__global__ void __launch_bounds__(kBlock)
grid_fill_kernel(long long* __restrict__ grid, double* __restrict__ best, size_t cells_total, long long sentinel)
{
    const double nan = __builtin_nan("");
    for (size_t t = (size_t)blockIdx.x * kBlock + threadIdx.x; t < cells_total; t += (size_t)gridDim.x * kBlock) {
        grid[t] = sentinel;
        best[t] = nan;
    }
}

// Filter flags folded into the header where the ranking kernel did not count them itself (option "fused" 0).
__global__ void __launch_bounds__(kBlock)
result_summary_kernel(ResultHeader* __restrict__ hdr, const uint8_t* __restrict__ flags, int n_cells)
{
    int nf = 0, na = 0;
    for (int c = threadIdx.x; c < n_cells; c += kBlock) {
        const uint8_t f = flags[c];
        nf += f != 0;
        na += (f & 4) != 0;
    }
    if (nf) atomicAdd(&hdr->n_flagged, (unsigned int)nf);
    if (na) atomicAdd(&hdr->n_band_alarm, (unsigned int)na);
}

// ---------------------------------------------------------------------------
// resolve: repeat penalty + argmin + exact tie-break, in the reference's order
// ---------------------------------------------------------------------------

struct ResolveArgs {
    const mm_cell* recs;
    int n_cells;
    int grid_w, grid_h;
    int L;
    int repeat_range, repeat_addition;
    int K;
    const int32_t* cand_index;
    const double* cand_score;
    long long* grid;            // [grid_h*grid_w], pre-filled with the sentinel L
    double* best_score;         // [grid_h*grid_w], pre-filled with NaN
    // pixel data for the exact tie-break
    const float* target; int rows, cols;
    CellResize cell_scale;      // how a detail pixel is made from the normal-size cell (cell_pixel)
    const uint8_t* masks;
    const float* lib;           // [variants][L][3][ppad], permuted pixel order per variant
    size_t lib_stride;          // floats between two variants
    const int* inv_perm;        // [variants][P]: plane position of original pixel p
    int var_of_mask[4];
    int size, P, ppad;
    double tie_band;            // relative gap under which two totals count as a near-tie
    unsigned long long* exact_count;
    double* scratch;            // [gridDim.x][L]: full score row of a cell whose list overflowed
    unsigned long long* dense_count;
    // wave resolve: rows precomputed by dense_rows_kernel for the marked cells
    // speculative short lists (engine.hip "speculate and verify")
    const uint8_t* deep;        // [n_cells] 1 = the list holds the full K the counting bound needs; null = all
    int k_fill;                 // entries filled in a non-deep (speculative) list
    uint8_t* insufficient;      // [n_cells] out: 1 = the pick of a non-deep cell could not be verified
    const int* row_slot;        // [n_cells] slot in `rows`, -1 = list is complete
    const double* dense_rows;   // [marked][L]
    // sweep resolve (resolve_wave_kernel mode 2): picks changed by this sweep
    unsigned int* sweep_changes;
    const unsigned int* prev_changes;   // (may be null) the counter of the sweep enqueued just before this one: 0 = the grid had settled, nothing to do
    unsigned int* insufficient_count;   // (may be null) unverified speculative picks of this launch, counted
};

// Repeat value of library image `cand` at padded cell (px, py):
// CPUPhotomosaicGenerator::calculateRepeats (CPUPhotomosaicGenerator.cpp:185-225)
// -- rows clamp(py-R,0,H)..py-1 over columns clamp(px-R,0,W)..clamp(px+R,0,W-1),
// then the same row's columns startX..px-1; untouched cells hold the sentinel
// and never match (the CUDA path's `bestFit[n] < L` test, PhotomosaicGenerator.cu:137).
__device__ __forceinline__ int repeat_value(const ResolveArgs& a, int px, int py, int cand)
{
    const int R = a.repeat_range;
    const int sy = min(max(py - R, 0), a.grid_h);
    const int sx = min(max(px - R, 0), a.grid_w);
    const int ex = min(max(px + R, 0), a.grid_w - 1);
    int rep = 0;
    for (int ry = sy; ry < py; ++ry)
        for (int rx = sx; rx <= ex; ++rx)
            if (a.grid[(size_t)ry * a.grid_w + rx] == cand) rep += a.repeat_addition;
    for (int rx = sx; rx < px; ++rx)
        if (a.grid[(size_t)py * a.grid_w + rx] == cand) rep += a.repeat_addition;
    return rep;
}

constexpr int kExactChunk = 2048;   // pixels per LDS chunk of the sequential re-sum

// The reference's own summation for ONE (cell, candidate): start from the repeat
// value as int -> double, add the active pixels' differences one by one in
// row-major order (CPUPhotomosaicGenerator.cpp:143-161).  All threads evaluate
// the per-pixel terms; thread 0 adds them in order.  Returns the total to every
// thread.
template <int TYPE>
__device__ double exact_total(const ResolveArgs& a, const mm_cell& rec, int cand, int rep,
                              double* buf, double* bcast)
{
    double running = (double)rep;
    const int v = a.var_of_mask[rec.mask_id];
    const float* l0 = a.lib + (size_t)v * a.lib_stride + (size_t)cand * 3 * a.ppad;
    const int* ip = a.inv_perm + (size_t)v * a.P;
    const uint8_t* mk = a.masks + (size_t)rec.mask_id * a.ppad;
    for (int base = 0; base < a.P; base += kExactChunk) {
        const int n = min(kExactChunk, a.P - base);
        for (int t = threadIdx.x; t < n; t += (int)blockDim.x) {
            const int p = base + t;
            const int row = p / a.size, col = p - row * a.size;
            double term = -1.0;                               // marker: not active
            if (mk[p] != 0 && row >= rec.row_min && row < rec.row_max &&
                col >= rec.col_min && col < rec.col_max) {
                float src[3];
                cell_pixel(a.target, a.rows, a.cols, rec.y, rec.x, row, col, a.cell_scale, src);
                const int k = ip[p];
                term = colour_diff<TYPE>(Px{src[0], src[1], src[2]},
                                         Px{l0[k], l0[a.ppad + k], l0[2 * (size_t)a.ppad + k]});
            }
            buf[t] = term;
        }
        __syncthreads();
        if (threadIdx.x == 0)
            for (int t = 0; t < n; ++t)
                if (buf[t] != -1.0) running += buf[t];
        __syncthreads();
    }
    if (threadIdx.x == 0) *bcast = running;
    __syncthreads();
    running = *bcast;
    __syncthreads();
    return running;
}

// Base score of EVERY library image for one cell, by one workgroup (slow path:
// only for cells whose candidate list was cut through a near-tie cluster).
template <int TYPE>
__device__ void dense_row(const ResolveArgs& a, const mm_cell& rec, double* row, double* sh4)
{
    const uint8_t* mk = a.masks + (size_t)rec.mask_id * a.ppad;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int v = a.var_of_mask[rec.mask_id];
    const int* ip = a.inv_perm + (size_t)v * a.P;
    {   // A cell with no active pixel (a sliver of an offset shape whose bound misses the mask) scores 0
        // against every image: the commonest reason for a whole row, and it needs no pixel pass.
        int any = 0;
        for (int p = threadIdx.x; p < a.P && !any; p += kBlock) {
            const int r = p / a.size, c = p - r * a.size;
            any = mk[p] != 0 && r >= rec.row_min && r < rec.row_max && c >= rec.col_min && c < rec.col_max;
        }
        if (!__syncthreads_or(any)) {
            for (int l = threadIdx.x; l < a.L; l += kBlock) row[l] = 0.0;
            __syncthreads();
            return;
        }
    }
    for (int l = 0; l < a.L; ++l) {
        const float* l0 = a.lib + (size_t)v * a.lib_stride + (size_t)l * 3 * a.ppad;
        double acc = 0.0;
        for (int p = threadIdx.x; p < a.P; p += kBlock) {
            const int r = p / a.size, c = p - r * a.size;
            if (mk[p] != 0 && r >= rec.row_min && r < rec.row_max && c >= rec.col_min && c < rec.col_max) {
                float src[3];
                cell_pixel(a.target, a.rows, a.cols, rec.y, rec.x, r, c, a.cell_scale, src);
                const int k = ip[p];
                acc += parallel_term<TYPE>(Px{src[0], src[1], src[2]},
                                            Px{l0[k], l0[a.ppad + k], l0[2 * (size_t)a.ppad + k]});
            }
        }
        acc = wave_sum(acc);
        __syncthreads();
        if (lane == 0) sh4[wave] = acc;
        __syncthreads();
        if (threadIdx.x == 0) row[l] = ((sh4[0] + sh4[1]) + sh4[2]) + sh4[3];
    }
    __syncthreads();
}

// sequential != 0: ONE workgroup walks the cells in raster order (each cell's
// penalty reads the cells chosen before it).  sequential == 0 (no repeats):
// workgroups take cells independently.
template <int TYPE>
__global__ void __launch_bounds__(kBlock)
resolve_kernel(ResolveArgs a, int sequential)
{
    __shared__ ScoreIdx sh[kBlock / kWave];
    __shared__ double buf[kExactChunk];
    __shared__ double bcast;
    __shared__ double sh4[kBlock / kWave];

    const int first = sequential ? 0 : blockIdx.x;
    const int stride = sequential ? 1 : gridDim.x;
    for (int c = first; c < a.n_cells; c += stride) {
        const mm_cell rec = a.recs[c];
        const int32_t* ci = a.cand_index + (size_t)c * a.K;
        const double* cs = a.cand_score + (size_t)c * a.K;
        const bool is_deep = !a.deep || a.deep[c] != 0;
        int K = is_deep ? a.K : min(a.K, a.k_fill);
        const bool penalise = a.repeat_range > 0 && a.repeat_addition != 0;
        if (is_deep && ci[a.K - 1] <= -2) {   // overflow mark: take the whole row instead
            double* row = a.scratch + (size_t)blockIdx.x * a.L;
            dense_row<TYPE>(a, rec, row, sh4);
            if (threadIdx.x == 0) atomicAdd(a.dense_count, 1ull);
            ci = nullptr;
            cs = row;
            K = a.L;
        }
        // a mark (<= -2) on a speculative list is just an entry
        auto cand_at = [&](int k) { const int v = ci ? ci[k] : k; return (!is_deep && v <= -2) ? -(v + 2) : v; };

        // pass 1: lowest (total, index) over the cell's candidates
        ScoreIdx best{kInf, 0x7fffffff};
        for (int k = threadIdx.x; k < K; k += kBlock) {
            const int cand = cand_at(k);
            if (cand < 0) continue;
            const int rep = penalise ? repeat_value(a, rec.grid_x, rec.grid_y, cand) : 0;
            const double total = (double)rep + cs[k];
            if (lex_less(total, cand, best.s, best.i)) { best.s = total; best.i = cand; }
        }
        best = block_min(best, sh);

        long long chosen = a.L;                // sentinel: "Failed to find a best fit" (:173-177)
        double chosen_total = best.s;
        if (best.i != 0x7fffffff) {
            chosen = best.i;
            // pass 2: how many candidates sit within the near-tie band of the minimum?
            const double thr = best.s + fabs(best.s) * a.tie_band;
            int near = 0, nonzero = 0;
            for (int k = threadIdx.x; k < K; k += kBlock) {
                const int cand = cand_at(k);
                if (cand < 0) continue;
                const int rep = penalise ? repeat_value(a, rec.grid_x, rec.grid_y, cand) : 0;
                if ((double)rep + cs[k] <= thr) { ++near; nonzero |= cs[k] != 0.0; }
            }
            const int many_in_one = __syncthreads_or(near > 1);
            const int holders = __syncthreads_count(near > 0);
            // base score exactly 0 => reference-order total is exactly (double)repeat: no re-sum
            const int any_nonzero = __syncthreads_or(nonzero);
            if ((many_in_one || holders > 1) && any_nonzero) {
                // Re-sum every contender in the reference's order, ascending index,
                // strict < : exactly the CPU scan restricted to the contenders.
                double ex_best = kInf;
                int ex_idx = -1;
                int prev = -1;
                for (;;) {
                    int nxt = 0x7fffffff;                // lowest contender index above prev
                    for (int k = threadIdx.x; k < K; k += kBlock) {
                        const int cand = cand_at(k);
                        if (cand <= prev || cand >= nxt) continue;
                        const int rep = penalise ? repeat_value(a, rec.grid_x, rec.grid_y, cand) : 0;
                        if ((double)rep + cs[k] <= thr) nxt = cand;
                    }
                    ScoreIdx key{(double)nxt, nxt};      // block-min on the index alone
                    key = block_min(key, sh);
                    if (key.i == 0x7fffffff) break;
                    const int nxt_rep = penalise ? repeat_value(a, rec.grid_x, rec.grid_y, key.i) : 0;
                    // base score of the winner index (whichever thread holds it tells the block)
                    int zero_here = 0;
                    for (int k = threadIdx.x; k < K; k += kBlock)
                        if (cand_at(k) == key.i && cs[k] == 0.0) zero_here = 1;
                    const bool is_zero = __syncthreads_or(zero_here) != 0;
                    const double ex = is_zero ? (double)nxt_rep : exact_total<TYPE>(a, rec, key.i, nxt_rep, buf, &bcast);
                    if (threadIdx.x == 0 && !is_zero) atomicAdd(a.exact_count, 1ull);
                    if (ex < ex_best) { ex_best = ex; ex_idx = key.i; }
                    prev = key.i;
                }
                if (ex_idx >= 0) { chosen = ex_idx; chosen_total = ex_best; }
            }
        }

        const size_t pos = (size_t)rec.grid_y * a.grid_w + rec.grid_x;
        if (threadIdx.x == 0) {
            a.grid[pos] = chosen;
            a.best_score[pos] = chosen_total;
            if (a.insufficient) {
                bool insuff = false;
                if (!is_deep) {           // see resolve_wave_kernel
                    int nvalid = 0;
                    while (nvalid < K && cand_at(nvalid) >= 0) ++nvalid;
                    const double b_last = nvalid > 0 ? cs[nvalid - 1] : -kInf;
                    insuff = !(chosen != a.L && chosen_total + fabs(chosen_total) * a.tie_band < b_last);
                }
                a.insufficient[c] = insuff ? 1 : 0;
            }
        }
        __syncthreads();          // workgroup-scope release/acquire: the next cell reads grid[]
    }
}

// ---------------------------------------------------------------------------
// resolve, wave form: one wave per cell chain, no workgroup barriers on the hot path
// ---------------------------------------------------------------------------

// Which cells carry the overflow mark, and a dense slot number for each of them.
__global__ void __launch_bounds__(kBlock)
mark_scan_kernel(const int32_t* __restrict__ cand_index, int n_cells, int K,
                 const uint8_t* __restrict__ deep, int* __restrict__ row_slot, int* __restrict__ count)
{
    // a mark on a speculative (non-deep) list is ignored: such a cell is verified by the
    // sufficiency test instead and deepened if that fails
    for (int c = blockIdx.x * kBlock + threadIdx.x; c < n_cells; c += gridDim.x * kBlock)
        row_slot[c] = ((!deep || deep[c]) && cand_index[(size_t)c * K + K - 1] <= -2) ? atomicAdd(count, 1) : -1;
}

// Full fp64 score rows of the marked cells, one workgroup per cell (parallel: base
// scores do not depend on the repeat penalty, so this never sits on the serial chain).
template <int TYPE>
__global__ void __launch_bounds__(kBlock)
dense_rows_kernel(ResolveArgs a, double* __restrict__ rows)
{
    __shared__ double sh4[kBlock / kWave];
    const int slot = a.row_slot[blockIdx.x];
    if (slot < 0) return;
    dense_row<TYPE>(a, a.recs[blockIdx.x], rows + (size_t)slot * a.L, sh4);
    if (threadIdx.x == 0) atomicAdd(a.dense_count, 1ull);
}

// sequential == 1: ONE wave walks the cells in raster order; the chosen indices of the
// last R+1 grid rows live in an LDS ring (the only state calculateRepeats reads,
// CPUPhotomosaicGenerator.cpp:189-224).  Lane k holds candidate k; the <= R(2R+1)+R
// neighbour picks are fetched once into lanes and matched against every lane's
// candidate with v_readlane broadcasts; a wave-wide (total, index) arg-min and a ballot
// for the near-tie test finish the cell.  The next cell's record and candidates are
// loaded while the current one is processed.  sequential == 0: waves take cells
// independently (no penalty, no ring).
//
// sequential == 2, the SWEEP form of the raster-order dependency: every cell is
// re-decided in parallel from the picks currently in a.grid (in place), and the number
// of cells whose pick changed is counted.  The reference's result is the unique fix
// point of "cell = f(picks of the cells before it in raster order)" (induction over the
// raster order), and a sweep that changes nothing saw one constant state, i.e. that
// fix point.  Cells only read cells earlier in raster order, so after s sweeps every
// cell whose dependency chain is shorter than s is final: the host launches sweeps
// until one reports no change (a handful when collisions are local) and falls back to
// the sequential forms if a bound is passed.  Time no longer grows with the grid height,
// which is what the multi-GPU step needs: its resolve covers the cells of ALL ranks.
template <int TYPE>
__global__ void __launch_bounds__(kWave)
resolve_wave_kernel(ResolveArgs a, int sequential)
{
    extern __shared__ int ring[];              // [(R+1)][grid_w], sequential == 1 only
    __shared__ double buf[kExactChunk];
    __shared__ double bcast;
    const int lane = threadIdx.x;
    const bool sweep = sequential == 2;
    const bool penalise = sequential && a.repeat_range > 0 && a.repeat_addition != 0;
    const int R = a.repeat_range, ring_rows = R + 1, gw = a.grid_w;
    const int sentinel = a.L;
    // A set of sweeps is enqueued without asking the host in between: once a sweep has changed nothing the grid is at its fix point
    // and the rest of the set returns at once (their own counters stay 0, as the block was initialised)
    if (sweep && a.prev_changes != nullptr && *a.prev_changes == 0u) return;
    if (sweep) sequential = 0;                 // cells are taken independently
    if (penalise && !sweep) {
        for (int i = lane; i < ring_rows * gw; i += kWave) ring[i] = sentinel;
        __syncthreads();
    }
    int cur_row = -1;

    const int first = sequential ? 0 : blockIdx.x;
    const int stride = sequential ? 1 : gridDim.x;
    if (first >= a.n_cells) return;

    // software pipeline: (rec, my candidate) of the cell about to be processed
    mm_cell rec = a.recs[first];
    int my_ci = lane < a.K ? a.cand_index[(size_t)first * a.K + lane] : -1;
    double my_cs = lane < a.K ? a.cand_score[(size_t)first * a.K + lane] : kInf;

    for (int c = first; c < a.n_cells; c += stride) {
        const int nc = c + stride;
        mm_cell nrec = rec;
        int nci = -1;
        double ncs = kInf;
        if (nc < a.n_cells) {                   // prefetch the next cell
            nrec = a.recs[nc];
            if (lane < a.K) { nci = a.cand_index[(size_t)nc * a.K + lane]; ncs = a.cand_score[(size_t)nc * a.K + lane]; }
        }

        const int px = rec.grid_x, py = rec.grid_y;
        int n_above = 0, wrow = 0, sx = 0, sy = 0, nn = 0;
        if (penalise) {
            if (!sweep && py != cur_row) {      // entering a grid row: recycle the ring rows it needs
                for (int r = max(cur_row + 1, py - R); r <= py; ++r)
                    for (int x = lane; x < gw; x += kWave) ring[(r % ring_rows) * gw + x] = sentinel;
                cur_row = py;
                __syncthreads();
            }
            sy = min(max(py - R, 0), a.grid_h);
            sx = min(max(px - R, 0), gw);
            const int ex = min(max(px + R, 0), gw - 1);
            wrow = ex - sx + 1;
            n_above = (py - sy) * wrow;
            nn = n_above + (px - sx);
        }

        // dense row instead of the list?
        const int slot = a.row_slot ? a.row_slot[c] : -1;
        const bool dense = slot >= 0;
        const bool is_deep = !a.deep || a.deep[c] != 0;
        const int K = dense ? a.L : (is_deep ? a.K : min(a.K, a.k_fill));
        const double* row = dense ? a.dense_rows + (size_t)slot * a.L : nullptr;

        // candidate of this lane in chunk `base`, with its total
        auto eval = [&](int base, int& cand, double& total, int& rep_out, double& base_out) {
            const int k = base + lane;
            double s;
            if (dense) { cand = k < K ? k : -1; s = k < K ? row[k] : kInf; }
            else if (base == 0) { cand = k < K ? my_ci : -1; s = k < K ? my_cs : kInf; }
            else { cand = k < K ? a.cand_index[(size_t)c * a.K + k] : -1; s = k < K ? a.cand_score[(size_t)c * a.K + k] : kInf; }
            if (!is_deep && cand <= -2) cand = -(cand + 2);     // a mark on a speculative list is just an entry
            int rep = 0;
            for (int t0 = 0; t0 < nn; t0 += kWave) {        // neighbours, 64 at a time
                const int t = t0 + lane;
                int nbv = sentinel;
                if (t < nn) {
                    int ry, rx;
                    if (t < n_above) { const int q = t / wrow; ry = sy + q; rx = sx + (t - q * wrow); }
                    else { ry = py; rx = sx + (t - n_above); }
                    nbv = sweep ? (int)__atomic_load_n(&a.grid[(size_t)ry * gw + rx], __ATOMIC_RELAXED)
                                : ring[(ry % ring_rows) * gw + rx];
                }
                const int cnt = min(kWave, nn - t0);
                for (int j = 0; j < cnt; ++j)
                    rep += (__builtin_amdgcn_readlane(nbv, j) == cand) ? a.repeat_addition : 0;
            }
            rep_out = rep;
            base_out = s;
            total = (double)rep + s;
        };

        // pass 1: lowest (total, index)
        ScoreIdx best{kInf, 0x7fffffff};
        int c0 = -1, r0 = 0;
        double t0v = kInf, s0 = kInf;
        for (int base = 0; base < K; base += kWave) {
            int cand, rep; double total, sb;
            eval(base, cand, total, rep, sb);
            if (base == 0) { c0 = cand; t0v = total; r0 = rep; s0 = sb; }
            if (cand >= 0 && lex_less(total, cand, best.s, best.i)) { best.s = total; best.i = cand; }
        }
        best = wave_min(best);

        long long chosen = a.L;                 // sentinel: "Failed to find a best fit" (:173-177)
        double chosen_total = best.s;
        if (best.i != 0x7fffffff) {
            chosen = best.i;
            const double thr = best.s + fabs(best.s) * a.tie_band;
            // A contender whose base score is exactly 0 (no active pixel, or every pixel
            // identical) has the reference-order total (double)repeat + 0 exactly: nothing
            // to re-sum.  Only contenders with a non-zero base need the exact pass.
            int near = 0, nonzero = 0;
            for (int base = 0; base < K; base += kWave) {
                int cand, rep; double total, sb;
                if (base == 0) { cand = c0; total = t0v; sb = s0; } else eval(base, cand, total, rep, sb);
                near += __popcll(__ballot(cand >= 0 && total <= thr));
                nonzero += __popcll(__ballot(cand >= 0 && total <= thr && sb != 0.0));
            }
            if (near > 1 && nonzero > 0) {
                // Re-sum every contender in the reference's order, ascending index, strict <
                double ex_best = kInf;
                int ex_idx = -1, prev = -1;
                for (;;) {
                    int key_i = 0x7fffffff, key_rep = 0, key_zero = 0;   // lowest contender index above prev
                    for (int base = 0; base < K; base += kWave) {
                        int cand, rep; double total, sb;
                        if (base == 0) { cand = c0; total = t0v; rep = r0; sb = s0; } else eval(base, cand, total, rep, sb);
                        if (cand > prev && total <= thr && cand < key_i) { key_i = cand; key_rep = rep; key_zero = sb == 0.0; }
                    }
                    const ScoreIdx win = wave_min(ScoreIdx{(double)key_i, key_i});
                    if (win.i == 0x7fffffff) break;
                    // the lane holding `win.i` knows its repeat value
                    const unsigned long long who = __ballot(key_i == win.i);
                    const int src = __ffsll((long long)who) - 1;
                    const int win_rep = __shfl(key_rep, src, kWave);
                    const bool win_zero = __shfl(key_zero, src, kWave) != 0;
                    const double ex = win_zero ? (double)win_rep : exact_total<TYPE>(a, rec, win.i, win_rep, buf, &bcast);
                    if (lane == 0 && !win_zero) atomicAdd(a.exact_count, 1ull);
                    if (ex < ex_best) { ex_best = ex; ex_idx = win.i; }
                    prev = win.i;
                }
                if (ex_idx >= 0) { chosen = ex_idx; chosen_total = ex_best; }
            }
        }

        if (a.insufficient) {
            // Speculative list: any unlisted candidate has a base score >= the list's last (largest)
            // one and a penalty >= 0, so the pick is certainly right when its total is below that.
            bool insuff = false;
            if (!is_deep) {
                const int nvalid = __popcll(__ballot(c0 >= 0));
                const double b_last = nvalid > 0 ? __shfl(s0, nvalid - 1, kWave) : -kInf;
                insuff = !(chosen != a.L && chosen_total + fabs(chosen_total) * a.tie_band < b_last);
            }
            if (lane == 0) {
                a.insufficient[c] = insuff ? 1 : 0;
                if (insuff && a.insufficient_count) atomicAdd(a.insufficient_count, 1u);
            }
        }
        if (lane == 0) {
            const size_t pos = (size_t)py * gw + px;
            if (sweep) {
                if (__atomic_load_n(&a.grid[pos], __ATOMIC_RELAXED) != chosen) atomicAdd(a.sweep_changes, 1u);
                __atomic_store_n(&a.grid[pos], chosen, __ATOMIC_RELAXED);
            } else {
                a.grid[pos] = chosen;
            }
            a.best_score[pos] = chosen_total;
            if (penalise && !sweep) ring[(py % ring_rows) * gw + px] = (int)chosen;
        }
        if (penalise && !sweep) __syncthreads();    // single wave: orders the LDS write before the next reads

        rec = nrec;
        my_ci = nci;
        my_cs = ncs;
    }
}

// ---------------------------------------------------------------------------
// resolve, row-pipelined form: the raster-order dependency as a wavefront
// ---------------------------------------------------------------------------
//
// calculateRepeats at (x, y) reads rows y-R..y-1 over columns x-R..x+R and row y to
// the left (CPUPhotomosaicGenerator.cpp:189-224).  So row y may work on column x as
// soon as row y-1 has finished column x+R: up to grid_w/(R+1) rows can be in flight.
// One workgroup, one wave per row in flight; the whole padded grid of chosen indices
// and one progress counter per row live in LDS.  progress[y] = p means every cell of
// row y with grid_x < p is final.  A wave spins (bounded, with s_sleep) on the row
// above; rows are taken in increasing order, so the wait graph is acyclic.

constexpr int kRowWaves = 8;
constexpr int kRowExactChunk = 128;      // doubles of per-wave LDS for the exact re-sum

// First cell and cell count of every grid row (cells are in raster order).
__global__ void __launch_bounds__(kBlock)
row_index_kernel(const mm_cell* __restrict__ recs, int n_cells, int* __restrict__ row_first,
                 int* __restrict__ row_count)
{
    for (int c = blockIdx.x * kBlock + threadIdx.x; c < n_cells; c += gridDim.x * kBlock) {
        const int y = recs[c].grid_y;
        if (c == 0 || recs[c - 1].grid_y != y) row_first[y] = c;
        atomicAdd(&row_count[y], 1);
    }
}

// exact_total for ONE wave (no workgroup barrier): same arithmetic, same order.
template <int TYPE>
__device__ double exact_total_wave(const ResolveArgs& a, const mm_cell& rec, int cand, int rep, double* wbuf)
{
    const int lane = threadIdx.x & 63;
    double running = (double)rep;
    const int v = a.var_of_mask[rec.mask_id];
    const float* l0 = a.lib + (size_t)v * a.lib_stride + (size_t)cand * 3 * a.ppad;
    const int* ip = a.inv_perm + (size_t)v * a.P;
    const uint8_t* mk = a.masks + (size_t)rec.mask_id * a.ppad;
    for (int base = 0; base < a.P; base += kRowExactChunk) {
        const int n = min(kRowExactChunk, a.P - base);
        for (int t = lane; t < n; t += kWave) {
            const int p = base + t;
            const int row = p / a.size, col = p - row * a.size;
            double term = -1.0;
            if (mk[p] != 0 && row >= rec.row_min && row < rec.row_max &&
                col >= rec.col_min && col < rec.col_max) {
                float src[3];
                cell_pixel(a.target, a.rows, a.cols, rec.y, rec.x, row, col, a.cell_scale, src);
                const int k = ip[p];
                term = colour_diff<TYPE>(Px{src[0], src[1], src[2]},
                                         Px{l0[k], l0[a.ppad + k], l0[2 * (size_t)a.ppad + k]});
            }
            wbuf[t] = term;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        if (lane == 0)
            for (int t = 0; t < n; ++t)
                if (wbuf[t] != -1.0) running += wbuf[t];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
    return __shfl(running, 0, kWave);
}

template <int TYPE>
__global__ void __launch_bounds__(kRowWaves * kWave)
resolve_rows_kernel(ResolveArgs a, const int* __restrict__ row_first, const int* __restrict__ row_count,
                    int* __restrict__ stuck)
{
    extern __shared__ int lds[];
    const int gw = a.grid_w, gh = a.grid_h;
    int* lgrid = lds;                                  // [gh][gw] chosen index, sentinel L
    volatile int* progress = lds + (size_t)gw * gh;    // [gh]
    volatile int* abort_flag = progress + gh;          // set once if a wait times out
    double* exact_buf = reinterpret_cast<double*>(lds + (((size_t)gw * gh + gh + 2) & ~(size_t)1));
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    double* wbuf = exact_buf + (size_t)wave * kRowExactChunk;
    const int R = a.repeat_range, sentinel = a.L;

    for (int i = threadIdx.x; i < gw * gh; i += blockDim.x) lgrid[i] = sentinel;
    for (int y = threadIdx.x; y < gh; y += blockDim.x) progress[y] = row_count[y] == 0 ? gw : 0;
    if (threadIdx.x == 0) *abort_flag = 0;
    __syncthreads();                                   // the only workgroup barrier

    for (int py = wave; py < gh; py += kRowWaves) {
        const int n_row = row_count[py];
        if (n_row == 0) continue;
        const int c_first = row_first[py];
        mm_cell rec = a.recs[c_first];
        int my_ci = lane < a.K ? a.cand_index[(size_t)c_first * a.K + lane] : -1;
        double my_cs = lane < a.K ? a.cand_score[(size_t)c_first * a.K + lane] : kInf;

        for (int i = 0; i < n_row; ++i) {
            const int c = c_first + i;
            mm_cell nrec = rec;
            int nci = -1;
            double ncs = kInf;
            if (i + 1 < n_row) {                       // prefetch the row's next cell
                nrec = a.recs[c + 1];
                if (lane < a.K) { nci = a.cand_index[(size_t)(c + 1) * a.K + lane]; ncs = a.cand_score[(size_t)(c + 1) * a.K + lane]; }
            }
            const int px = rec.grid_x;
            const int sy = min(max(py - R, 0), gh);
            const int sx = min(max(px - R, 0), gw);
            const int ex = min(max(px + R, 0), gw - 1);
            const int wrow = ex - sx + 1;
            const int n_above = (py - sy) * wrow;
            const int nn = n_above + (px - sx);

            // Wait until every row of the window has cleared columns <= ex.  (Each row is
            // checked: with ragged rows -- offset cell shapes -- "row y-1 is past ex" does
            // not imply the rows above it are.)  The spin is bounded; a timeout raises
            // `stuck` and releases every wave, so the kernel always drains.
            for (int ry = sy; ry < py; ++ry) {
                int spins = 0;
                while (progress[ry] < ex + 1 && *abort_flag == 0) {
                    __builtin_amdgcn_s_sleep(2);
                    if (++spins > (1 << 22)) { *abort_flag = 1; if (lane == 0) atomicExch(stuck, 1); }
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");

            const int slot = a.row_slot ? a.row_slot[c] : -1;
            const bool dense = slot >= 0;
            const bool is_deep = !a.deep || a.deep[c] != 0;
            const int K = dense ? a.L : (is_deep ? a.K : min(a.K, a.k_fill));
            const double* row = dense ? a.dense_rows + (size_t)slot * a.L : nullptr;

            auto eval = [&](int base, int& cand, double& total, int& rep_out, double& base_out) {
                const int k = base + lane;
                double s;
                if (dense) { cand = k < K ? k : -1; s = k < K ? row[k] : kInf; }
                else if (base == 0) { cand = k < K ? my_ci : -1; s = k < K ? my_cs : kInf; }
                else { cand = k < K ? a.cand_index[(size_t)c * a.K + k] : -1; s = k < K ? a.cand_score[(size_t)c * a.K + k] : kInf; }
                if (!is_deep && cand <= -2) cand = -(cand + 2);     // a mark on a speculative list is just an entry
                int rep = 0;
                for (int t0 = 0; t0 < nn; t0 += kWave) {
                    const int t = t0 + lane;
                    int nbv = sentinel;
                    if (t < nn) {
                        int ry, rx;
                        if (t < n_above) { const int q = t / wrow; ry = sy + q; rx = sx + (t - q * wrow); }
                        else { ry = py; rx = sx + (t - n_above); }
                        nbv = lgrid[ry * gw + rx];
                    }
                    const int cnt = min(kWave, nn - t0);
                    for (int j = 0; j < cnt; ++j)
                        rep += (__builtin_amdgcn_readlane(nbv, j) == cand) ? a.repeat_addition : 0;
                }
                rep_out = rep;
                base_out = s;
                total = (double)rep + s;
            };

            ScoreIdx best{kInf, 0x7fffffff};
            int c0 = -1, r0 = 0;
            double t0v = kInf, s0 = kInf;
            for (int base = 0; base < K; base += kWave) {
                int cand, rep; double total, sb;
                eval(base, cand, total, rep, sb);
                if (base == 0) { c0 = cand; t0v = total; r0 = rep; s0 = sb; }
                if (cand >= 0 && lex_less(total, cand, best.s, best.i)) { best.s = total; best.i = cand; }
            }
            best = wave_min(best);

            long long chosen = a.L;
            double chosen_total = best.s;
            if (best.i != 0x7fffffff) {
                chosen = best.i;
                const double thr = best.s + fabs(best.s) * a.tie_band;
                // A contender whose base score is exactly 0 (no active pixel, or every pixel
                // identical) has the reference-order total (double)repeat + 0 exactly: nothing
                // to re-sum.  Only contenders with a non-zero base need the exact pass.
                int near = 0, nonzero = 0;
                for (int base = 0; base < K; base += kWave) {
                    int cand, rep; double total, sb;
                    if (base == 0) { cand = c0; total = t0v; sb = s0; } else eval(base, cand, total, rep, sb);
                    near += __popcll(__ballot(cand >= 0 && total <= thr));
                    nonzero += __popcll(__ballot(cand >= 0 && total <= thr && sb != 0.0));
                }
                if (near > 1 && nonzero > 0) {
                    double ex_best = kInf;
                    int ex_idx = -1, prev = -1;
                    for (;;) {
                        int key_i = 0x7fffffff, key_rep = 0, key_zero = 0;
                        for (int base = 0; base < K; base += kWave) {
                            int cand, rep; double total, sb;
                            if (base == 0) { cand = c0; total = t0v; rep = r0; sb = s0; } else eval(base, cand, total, rep, sb);
                            if (cand > prev && total <= thr && cand < key_i) { key_i = cand; key_rep = rep; key_zero = sb == 0.0; }
                        }
                        const ScoreIdx win = wave_min(ScoreIdx{(double)key_i, key_i});
                        if (win.i == 0x7fffffff) break;
                        const unsigned long long who = __ballot(key_i == win.i);
                        const int src = __ffsll((long long)who) - 1;
                        const int win_rep = __shfl(key_rep, src, kWave);
                        const bool win_zero = __shfl(key_zero, src, kWave) != 0;
                        const double ex = win_zero ? (double)win_rep : exact_total_wave<TYPE>(a, rec, win.i, win_rep, wbuf);
                        if (lane == 0 && !win_zero) atomicAdd(a.exact_count, 1ull);
                        if (ex < ex_best) { ex_best = ex; ex_idx = win.i; }
                        prev = win.i;
                    }
                    if (ex_idx >= 0) { chosen = ex_idx; chosen_total = ex_best; }
                }
            }

            if (a.insufficient) {
                bool insuff = false;
                if (!is_deep) {
                    const int nvalid = __popcll(__ballot(c0 >= 0));
                    const double b_last = nvalid > 0 ? __shfl(s0, nvalid - 1, kWave) : -kInf;
                    insuff = !(chosen != a.L && chosen_total + fabs(chosen_total) * a.tie_band < b_last);
                }
                if (lane == 0) a.insufficient[c] = insuff ? 1 : 0;
            }
            if (lane == 0) {
                const size_t pos = (size_t)py * gw + px;
                a.grid[pos] = chosen;
                a.best_score[pos] = chosen_total;
                lgrid[pos] = (int)chosen;
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                // cells between this one and the row's next valid cell are not in the grid: final too
                progress[py] = (i + 1 < n_row) ? nrec.grid_x : gw;
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");    // our own next read of lgrid[py][px]

            rec = nrec;
            my_ci = nci;
            my_cs = ncs;
        }
    }
}

// ---------------------------------------------------------------------------
// resolve, FRONT form: the same wavefront, built for latency (round 6)
// ---------------------------------------------------------------------------
//
// The sweeps (resolve_wave_kernel mode 2) settle config D's synthetic content in four passes; on photographs neighbouring cells
// want the same few images, a change travels one cell per sweep along the raster order, and the grid of bench.py --config PD had
// not settled after 240 sweeps (4.3 ms of 18-us launches, then the fallback to resolve_rows_kernel: 2.4 ms -- twice per generate).
// The wavefront itself is exact in ONE pass; what made resolve_rows_kernel slow is its critical path: per cell two dependent
// global loads after the wait (row_slot, deep), a workgroup-scope release fence that drains the cell's GLOBAL result stores
// (a store's round trip, 1 - 2 us) before the row's progress may move, 128-cycle sleeps in the poll, and eight rows in flight.
// Here: everything that does not depend on the neighbours is loaded a cell ahead, the pick is published through LDS alone -- LDS operations of a wave complete in order, so `lgrid` then
// `progress` needs a wait on the LDS counter only -- and the global stores of a cell's results trail behind, off the path.
// Same per-cell arithmetic and tie handling as the other forms (a copy of resolve_rows_kernel's evaluation), same arguments.
// Eight waves, as resolve_rows_kernel: the cell loop with the fp64 reference form of the near-tie re-sum inlined wants ~200 VGPRs, and a
// 16-wave workgroup (128 per lane) spilled 60 of them into the loop (868 bytes of scratch per lane on the critical path of every cell).
constexpr int kFrontWaves = 8;

template <int TYPE>
__global__ void __launch_bounds__(kFrontWaves * kWave)
resolve_front_kernel(ResolveArgs a, const int* __restrict__ row_first, const int* __restrict__ row_count,
                     int* __restrict__ stuck)
{
    extern __shared__ int lds[];
    const int gw = a.grid_w, gh = a.grid_h;
    int* lgrid = lds;                                  // [gh][gw] chosen index, sentinel L
    int* progress = lds + (size_t)gw * gh;             // [gh]
    double* exact_buf = reinterpret_cast<double*>(lds + (((size_t)gw * gh + gh + 2) & ~(size_t)1));
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    double* wbuf = exact_buf + (size_t)wave * kRowExactChunk;
    // a row's results wait in LDS until the row is done: a global store inside the cell loop is waited for at the loop's next
    // vmcnt(0) (CDNA counts stores in vmcnt, and the prefetch of the next cell's list ends in one) -- a store's round trip per cell
    double* row_score = exact_buf + (size_t)kFrontWaves * kRowExactChunk + (size_t)wave * gw;     // [gw] totals of this wave's row
    unsigned char* row_flag = reinterpret_cast<unsigned char*>(exact_buf + (size_t)kFrontWaves * (kRowExactChunk + gw)) + (size_t)wave * gw;
    const int R = a.repeat_range, sentinel = a.L;

    for (int i = threadIdx.x; i < gw * gh; i += blockDim.x) lgrid[i] = sentinel;
    for (int y = threadIdx.x; y < gh; y += blockDim.x) progress[y] = row_count[y] == 0 ? gw : 0;
    __syncthreads();                                   // the only workgroup barrier

    // FAST PATH (the common shape: the window's <= R (2R + 1) + R neighbours fit one wave -- R <= 5 -- and so does the list):
    // lane t owns one neighbour slot for the whole kernel -- (dq, dr) = its offset from the cell, doff the same as a distance in the LDS
    // grid -- so a cell's neighbour fetch is an add, three range tests and ONE ds_read (no division by the clipped window's width);
    // the penalty is counted per CANDIDATE when the list is shorter than the window (ballot + popcount + a select per candidate:
    // 16 candidates against 40 neighbours); the arg-min runs on DPP rotations instead of 18 ds_bpermute.  Anything else -- a dense
    // row, a near-tie that needs the reference-order re-sum -- takes the general code below, which recomputes the cell from scratch.
    const int W = 2 * R + 1, n_slots = R * W + R;
    const bool fast_shape = n_slots <= kWave && a.K <= kWave && R > 0;
    int dq = 0, dr = 0;
    bool slot_used = false;
    if (fast_shape && lane < n_slots) {
        slot_used = true;
        if (lane < R * W) { const int q = lane / W; dq = q - R; dr = lane - q * W - R; }
        else { dq = 0; dr = lane - R * W - R; }
    }
    const int doff = dq * gw + dr;

    for (int py = wave; py < gh; py += kFrontWaves) {
        const int n_row = row_count[py];
        if (n_row == 0) continue;
        const int c_first = row_first[py];
        mm_cell rec = a.recs[c_first];
        int my_ci = lane < a.K ? a.cand_index[(size_t)c_first * a.K + lane] : -1;
        double my_cs = lane < a.K ? a.cand_score[(size_t)c_first * a.K + lane] : kInf;
        int my_slot = a.row_slot ? a.row_slot[c_first] : -1;
        int my_deep = a.deep ? (int)a.deep[c_first] : 1;

        for (int i = 0; i < n_row; ++i) {
            const int c = c_first + i;
            mm_cell nrec = rec;
            int nci = -1, nslot = -1, ndeep = 1;
            double ncs = kInf;
            if (i + 1 < n_row) {                       // everything of the row's next cell that does not depend on a neighbour
                nrec = a.recs[c + 1];
                if (lane < a.K) { nci = a.cand_index[(size_t)(c + 1) * a.K + lane]; ncs = a.cand_score[(size_t)(c + 1) * a.K + lane]; }
                if (a.row_slot) nslot = a.row_slot[c + 1];
                if (a.deep) ndeep = (int)a.deep[c + 1];
            }
            const int px = rec.grid_x;
            // cells between the row's previous valid cell and this one are not in the grid: final too
            if (lane == 0 && i > 0) __atomic_store_n(&progress[py], px, __ATOMIC_RELAXED);
            const int sy = min(max(py - R, 0), gh);
            const int sx = min(max(px - R, 0), gw);
            const int ex = min(max(px + R, 0), gw - 1);
            const int wrow = ex - sx + 1;
            const int n_above = (py - sy) * wrow;
            const int nn = n_above + (px - sx);

            // Wait until every row of the window has cleared columns <= ex (each row is checked: with ragged rows -- offset cell shapes --
            // "row y-1 is past ex" does not imply the rows above it are).  ONE LDS read per poll: lane r reads the progress word of window
            // row r.  Bounded: a timeout raises `stuck` and opens every row's progress, so every wave runs out and the kernel drains (the
            // host then reports the error; the grid of such a launch is not used).
            {
                const int n_rows_above = py - sy;
                int spins = 0;
                for (;;) {
                    int pr = 0x7fffffff;
                    for (int r0 = lane; r0 < n_rows_above; r0 += kWave) pr = min(pr, __atomic_load_n(&progress[sy + r0], __ATOMIC_RELAXED));
                    if (__ballot(pr < ex + 1) == 0ull) break;
                    __builtin_amdgcn_s_sleep(1);
                    if (++spins > (1 << 24)) {
                        for (int y = lane; y < gh; y += kWave) __atomic_store_n(&progress[y], 0x7ffffffe, __ATOMIC_RELAXED);
                        if (lane == 0) atomicExch(stuck, 1);
                        break;
                    }
                }
            }
            // (LDS only: the picks this cell reads were written before the progress words it has just seen, by in-order LDS writes)
            asm volatile("" ::: "memory");

            const int slot = my_slot;
            const bool dense = slot >= 0;
            const bool is_deep = my_deep != 0;
            const int K = dense ? a.L : (is_deep ? a.K : min(a.K, a.k_fill));
            const double* row = dense ? a.dense_rows + (size_t)slot * a.L : nullptr;

            long long chosen = a.L;                 // sentinel: "Failed to find a best fit" (CPUPhotomosaicGenerator.cpp:173-177)
            double chosen_total = kInf;
            int c0 = -1;
            double s0 = kInf;
            bool done = false;
            if (fast_shape && !dense) {
                const int cand = lane < K ? ((!is_deep && my_ci <= -2) ? -(my_ci + 2) : my_ci) : -1;
                const double sb = lane < K ? my_cs : kInf;
                const int ry = py + dq, rx = px + dr;
                int nbv = sentinel;
                if (slot_used && ry >= 0 && rx >= 0 && rx < gw) nbv = __atomic_load_n(&lgrid[py * gw + px + doff], __ATOMIC_RELAXED);
                int cnt = 0;
                // (a list is filled from its front and padded with empty entries behind: only the filled ones are looked up -- a short list in
                // a row read at the middle length of the deepening costs its own length, not the row's)
                const int n_listed = __popcll(__ballot(cand >= 0));
                if (n_listed <= n_slots) {
                    for (int k = 0; k < n_listed; ++k) {   // per candidate: how many neighbours hold it
                        const int ck = __builtin_amdgcn_readlane(cand, k);
                        const int hits = __popcll(__ballot(nbv == ck));
                        cnt = lane == k ? hits : cnt;
                    }
                } else {
                    for (int t = 0; t < n_slots; ++t) cnt += (__builtin_amdgcn_readlane(nbv, t) == cand) ? 1 : 0;
                }
                const int rep = cnt * a.repeat_addition;
                const double total = (double)rep + sb;
                ScoreIdx best{kInf, 0x7fffffff};
                if (cand >= 0) { best.s = total; best.i = cand; }
                best = wave_min_dpp(best);
                c0 = cand; s0 = sb;
                if (best.i == 0x7fffffff) {
                    done = true;                    // no candidate at all: the sentinel, total +inf (as the general code leaves it)
                } else {
                    const double thr = best.s + fabs(best.s) * a.tie_band;
                    const int near = __popcll(__ballot(cand >= 0 && total <= thr));
                    const int nonzero = __popcll(__ballot(cand >= 0 && total <= thr && sb != 0.0));
                    if (!(near > 1 && nonzero > 0)) { chosen = best.i; chosen_total = best.s; done = true; }
                }
            }
            if (!done) {

            auto eval = [&](int base, int& cand, double& total, int& rep_out, double& base_out) {
                const int k = base + lane;
                double s;
                if (dense) { cand = k < K ? k : -1; s = k < K ? row[k] : kInf; }
                else if (base == 0) { cand = k < K ? my_ci : -1; s = k < K ? my_cs : kInf; }
                else { cand = k < K ? a.cand_index[(size_t)c * a.K + k] : -1; s = k < K ? a.cand_score[(size_t)c * a.K + k] : kInf; }
                if (!is_deep && cand <= -2) cand = -(cand + 2);     // a mark on a speculative list is just an entry
                int rep = 0;
                for (int t0 = 0; t0 < nn; t0 += kWave) {
                    const int t = t0 + lane;
                    int nbv = sentinel;
                    if (t < nn) {
                        int ry, rx;
                        if (t < n_above) { const int q = t / wrow; ry = sy + q; rx = sx + (t - q * wrow); }
                        else { ry = py; rx = sx + (t - n_above); }
                        nbv = __atomic_load_n(&lgrid[ry * gw + rx], __ATOMIC_RELAXED);
                    }
                    // (lanes past the window hold the sentinel, which no candidate equals: whole octets, no tail test)
                    const int cnt = min(kWave, nn - t0);
                    for (int j = 0; j < cnt; j += 8) {
#pragma unroll
                        for (int u = 0; u < 8; ++u) rep += (__builtin_amdgcn_readlane(nbv, j + u) == cand) ? 1 : 0;
                    }
                }
                rep *= a.repeat_addition;
                rep_out = rep;
                base_out = s;
                total = (double)rep + s;
            };

            ScoreIdx best{kInf, 0x7fffffff};
            int r0 = 0;
            double t0v = kInf;
            for (int base = 0; base < K; base += kWave) {
                int cand, rep; double total, sb;
                eval(base, cand, total, rep, sb);
                if (base == 0) { c0 = cand; t0v = total; r0 = rep; s0 = sb; }
                if (cand >= 0 && lex_less(total, cand, best.s, best.i)) { best.s = total; best.i = cand; }
            }
            best = wave_min(best);

            chosen = a.L;
            chosen_total = best.s;
            if (best.i != 0x7fffffff) {
                chosen = best.i;
                const double thr = best.s + fabs(best.s) * a.tie_band;
                int near = 0, nonzero = 0;
                for (int base = 0; base < K; base += kWave) {
                    int cand, rep; double total, sb;
                    if (base == 0) { cand = c0; total = t0v; sb = s0; } else eval(base, cand, total, rep, sb);
                    near += __popcll(__ballot(cand >= 0 && total <= thr));
                    nonzero += __popcll(__ballot(cand >= 0 && total <= thr && sb != 0.0));
                }
                if (near > 1 && nonzero > 0) {         // re-sum every contender in the reference's order, ascending index, strict <
                    double ex_best = kInf;
                    int ex_idx = -1, prev = -1;
                    for (;;) {
                        int key_i = 0x7fffffff, key_rep = 0, key_zero = 0;
                        for (int base = 0; base < K; base += kWave) {
                            int cand, rep; double total, sb;
                            if (base == 0) { cand = c0; total = t0v; rep = r0; sb = s0; } else eval(base, cand, total, rep, sb);
                            if (cand > prev && total <= thr && cand < key_i) { key_i = cand; key_rep = rep; key_zero = sb == 0.0; }
                        }
                        const ScoreIdx win = wave_min(ScoreIdx{(double)key_i, key_i});
                        if (win.i == 0x7fffffff) break;
                        const unsigned long long who = __ballot(key_i == win.i);
                        const int src = __ffsll((long long)who) - 1;
                        const int win_rep = __shfl(key_rep, src, kWave);
                        const bool win_zero = __shfl(key_zero, src, kWave) != 0;
                        const double ex = win_zero ? (double)win_rep : exact_total_wave<TYPE>(a, rec, win.i, win_rep, wbuf);
                        if (lane == 0 && !win_zero) atomicAdd(a.exact_count, 1ull);
                        if (ex < ex_best) { ex_best = ex; ex_idx = win.i; }
                        prev = win.i;
                    }
                    if (ex_idx >= 0) { chosen = ex_idx; chosen_total = ex_best; }
                }
            }
            }   // general path

            // publish the pick: LDS only, in order (the grid entry, then the row's progress)
            if (lane == 0) {
                const size_t pos = (size_t)py * gw + px;
                __atomic_store_n(&lgrid[pos], (int)chosen, __ATOMIC_RELAXED);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                // (up to and including this cell; the gap to the row's next valid cell -- ragged rows -- is added when that cell's turn
                // comes, so that nothing here needs the NEXT record yet: asking for nrec.grid_x at this point made the compiler wait for
                // the prefetch right after issuing it, a global load's latency on every cell's path)
                __atomic_store_n(&progress[py], (i + 1 < n_row) ? px + 1 : gw, __ATOMIC_RELAXED);
            }
            // ... and the cell's results into the row's LDS staging, off the critical path
            bool insuff = false;
            if (a.insufficient && !is_deep) {
                const int nvalid = __popcll(__ballot(c0 >= 0));
                double b_last = -kInf;
                if (nvalid > 0) {                   // (a uniform lane number: two v_readlane, no LDS crossbar)
                    const unsigned long long bits = __builtin_bit_cast(unsigned long long, s0);
                    const unsigned int lo = (unsigned int)__builtin_amdgcn_readlane((int)(unsigned int)bits, nvalid - 1);
                    const unsigned int hi = (unsigned int)__builtin_amdgcn_readlane((int)(unsigned int)(bits >> 32), nvalid - 1);
                    b_last = __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
                }
                insuff = !(chosen != a.L && chosen_total + fabs(chosen_total) * a.tie_band < b_last);
            }
            if (lane == 0) {
                row_score[i] = chosen_total;
                row_flag[i] = insuff ? 1 : 0;
            }
            asm volatile("" ::: "memory");             // (our own next read of lgrid[py][px] comes after its write in program order)

            rec = nrec;
            my_ci = nci;
            my_cs = ncs;
            my_slot = nslot;
            my_deep = ndeep;
        }
        // the row's results to global memory, all lanes (cells of a row are consecutive records)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        unsigned int n_insuff = 0;
        for (int i = lane; i < n_row; i += kWave) {
            const size_t pos = (size_t)py * gw + a.recs[c_first + i].grid_x;
            a.grid[pos] = (long long)__atomic_load_n(&lgrid[pos], __ATOMIC_RELAXED);
            a.best_score[pos] = row_score[i];
            if (a.insufficient) { a.insufficient[c_first + i] = row_flag[i]; n_insuff += row_flag[i]; }
        }
        if (a.insufficient_count) {
            for (int off = 32; off > 0; off >>= 1) n_insuff += __shfl_xor(n_insuff, off, kWave);
            if (lane == 0 && n_insuff != 0u) atomicAdd(a.insufficient_count, n_insuff);
        }
    }
}

// ---------------------------------------------------------------------------
// buildPhotomosaic (PhotomosaicGeneratorBase.cpp:110-207), one size step
// ---------------------------------------------------------------------------
//
// The reference paints cells in raster order through their flipped masks
// (cv::Mat::copyTo with mask, :176-178): where shapes overlap, the LATER cell wins.
// On the device that painter's order becomes an owner map: pass 1 raises
// owner[pixel] to (cell order + 1) with atomicMax for every masked in-bounds pixel of
// every placed cell (order-independent, exact); pass 2 reads one owner per output
// pixel and copies that cell's library pixel (+ alpha 255) or the background.
// Pure byte / index work: HBM-bound, 4 B owner + 4 B output per pixel.

__global__ void __launch_bounds__(kBlock)
mosaic_owner_kernel(const mm_cell* __restrict__ recs, const long long* __restrict__ best_fit,
                    int grid_w, int L, const uint8_t* __restrict__ masks, int size, int P, int ppad,
                    int cols, unsigned int* __restrict__ owner)
{
    const mm_cell rec = recs[blockIdx.y];
    const long long fit = best_fit[(size_t)rec.grid_y * grid_w + rec.grid_x];
    if (fit < 0 || fit >= L) return;                       // cell without a value: skipped (:147-150)
    const int p = blockIdx.x * kBlock + threadIdx.x;
    if (p >= P) return;
    const int row = p / size, col = p - row * size;
    if (row < rec.row_min || row >= rec.row_max || col < rec.col_min || col >= rec.col_max) return;
    if (masks[(size_t)rec.mask_id * ppad + p] == 0) return;
    atomicMax(&owner[(size_t)(rec.y + row) * cols + (rec.x + col)], (unsigned int)blockIdx.y + 1u);
}

// covered: u8 per pixel, 1 = a cell mask of this or an earlier size step holds the pixel
// (mosaicMask, PhotomosaicGeneratorBase.cpp:115,182-184).  first != 0: size step 0, every pixel
// is written (:197-201); otherwise only pixels no earlier step covers (:189-196).
__global__ void __launch_bounds__(kBlock)
mosaic_compose_kernel(const unsigned int* __restrict__ owner, const mm_cell* __restrict__ recs,
                      const long long* __restrict__ best_fit, int grid_w,
                      const uint8_t* __restrict__ lib_bgr, int size, int cols, size_t n_pixels,
                      uint32_t background, uint32_t* __restrict__ mosaic,
                      uint8_t* __restrict__ covered, int first)
{
    for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n_pixels; i += (size_t)gridDim.x * kBlock) {
        if (!first && covered[i]) continue;
        const unsigned int o = owner[i];
        uint32_t px = background;
        if (o != 0) {
            const mm_cell rec = recs[o - 1];
            const int y = (int)(i / cols), x = (int)(i - (size_t)y * cols);
            const long long fit = best_fit[(size_t)rec.grid_y * grid_w + rec.grid_x];
            const uint8_t* src = lib_bgr + ((size_t)fit * size * size + (size_t)(y - rec.y) * size + (x - rec.x)) * 3;
            px = (uint32_t)src[0] | ((uint32_t)src[1] << 8) | ((uint32_t)src[2] << 16) | 0xff000000u;
        }
        mosaic[i] = px;
        covered[i] = o != 0;
    }
}

// ImageUtility::batchResizeMat on the u8 library between the size steps of buildPhotomosaic
// (PhotomosaicGeneratorBase.cpp:126-127): INTER_AREA by a whole factor, the u8 rule of
// preprocess_u8_kernel (the alpha plane the reference carries is a constant 255).
__global__ void __launch_bounds__(kBlock)
shrink_u8_kernel(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst, int n_images, int size_in, int f)
{
    const int so = size_in / f;
    const size_t total = (size_t)n_images * so * so;
    const float scale = 1.0f / (float)(f * f);
    for (size_t t = (size_t)blockIdx.x * kBlock + threadIdx.x; t < total; t += (size_t)gridDim.x * kBlock) {
        const size_t img = t / ((size_t)so * so);
        const int p = (int)(t - img * so * so);
        const int r = p / so, c = p - r * so;
        const uint8_t* s = src + img * (size_t)size_in * size_in * 3;
        int sum[3] = {0, 0, 0};
        for (int i = 0; i < f; ++i)
            for (int j = 0; j < f; ++j) {
                const uint8_t* q = s + ((size_t)(r * f + i) * size_in + (c * f + j)) * 3;
                sum[0] += q[0]; sum[1] += q[1]; sum[2] += q[2];
            }
        for (int k = 0; k < 3; ++k)
            dst[t * 3 + k] = (uint8_t)(f == 2 ? (sum[k] + 2) >> 2 : min(255, max(0, __float2int_rn((float)sum[k] * scale))));
    }
}

// ---------------------------------------------------------------------------
// entropy subdivision of the grid (GridGenerator::findCellState, Grid/GridGenerator.cpp:143-189)
// ---------------------------------------------------------------------------
//
// A cell of a size step that is not the last is handed to the next (half-size) step when the grey-level
// entropy of the image under its (bounded, flipped) detail mask reaches 0.7 * MAX_ENTROPY = 5.6 bits:
//   visible part of the u8 main image under the cell rect  ->  resizeImage(EXACT) to the bounded mask's size
//   (Other/ImageUtility.cpp:34-61: INTER_AREA for a shrink)  ->  BGR2GRAY  ->  256-bin histogram under the mask
//   ->  entropy = -sum p log2 p  (Other/ImageUtility.cpp:189-242).
// One workgroup per cell: every thread resizes its share of the destination pixels straight from the resident
// image, the histogram lives in LDS, thread 0 sums the 256 terms in index order in fp64.
// OpenCV is absent from the reference checkout, so the resize arithmetic is the SPECIFIED one of this repository
// (imageutil.resize_area_u8; the CPU checker of the test suite restates it independently): same size -> copy; both axes shrunk by the same
// whole factor f -> (sum + 2) >> 2 for f = 2, rint(float(sum) * (1.f / (f f))) above; anything else ->
// computeResizeAreaTab per axis (entries generated here on the fly, in fp64 as cv::resize does) with f32
// accumulators: buf = ((0 + S a0) + S a1) + ... along each contributing source row, sum = b0 buf0, sum += b1 buf1 ...,
// cvRound, saturate.  BGR2GRAY on u8 is (B 1868 + G 9617 + R 4899 + 2^13) >> 14.

// Source cells and weights under destination index d of an axis shrunk from ssize to dsize.
struct AreaSpan {
    int first, count;        // source indices first .. first + count - 1
    float head, mid, tail;   // weights: entry 0 (if has_head), the full ones, the last (if has_tail)
    bool has_head, has_tail;
    __device__ __forceinline__ float weight(int k) const
    {
        if (has_head && k == 0) return head;
        if (has_tail && k == count - 1) return tail;
        return mid;
    }
};

__device__ __forceinline__ AreaSpan area_span(int ssize, int dsize, int d)
{
    const double scale = 1.0 / ((double)dsize / ssize);
    const double fsx1 = d * scale, fsx2 = fsx1 + scale;
    const double cell = scale < ssize - fsx1 ? scale : ssize - fsx1;
    int sx1 = (int)ceil(fsx1), sx2 = (int)floor(fsx2);
    if (sx2 > ssize - 1) sx2 = ssize - 1;
    if (sx1 > sx2) sx1 = sx2;
    AreaSpan sp;
    sp.has_head = sx1 - fsx1 > 1e-3;
    sp.has_tail = fsx2 - sx2 > 1e-3;
    sp.first = sp.has_head ? sx1 - 1 : sx1;
    sp.count = (sp.has_head ? 1 : 0) + (sx2 - sx1) + (sp.has_tail ? 1 : 0);
    sp.head = (float)((sx1 - fsx1) / cell);
    sp.mid = (float)(1.0 / cell);
    double w = fsx2 - sx2;
    if (w > 1.0) w = 1.0;
    if (w > cell) w = cell;
    sp.tail = (float)(w / cell);
    return sp;
}

__global__ void __launch_bounds__(kBlock)
cell_entropy_kernel(const mm_cell* __restrict__ recs, int n_cells, const uint8_t* __restrict__ image, int rows, int cols,
                    int normal_size, const uint8_t* __restrict__ masks, int detail_size, int ppad,
                    uint8_t* __restrict__ exceeded, double* __restrict__ entropy_out)
{
    __shared__ unsigned int hist[256];
    const int cell = blockIdx.x;
    if (cell >= n_cells) return;
    const mm_cell rec = recs[cell];
    hist[threadIdx.x] = 0u;                                   // kBlock == 256 bins
    __syncthreads();
    // visible part of the image under the normal cell's rect, and the bounded detail mask it is resized to
    const int y0 = min(max(rec.y, 0), rows), y1 = min(max(rec.y + normal_size, 0), rows);
    const int x0 = min(max(rec.x, 0), cols), x1 = min(max(rec.x + normal_size, 0), cols);
    const int vr = y1 - y0, vc = x1 - x0;
    const int bh = rec.row_max - rec.row_min, bw = rec.col_max - rec.col_min;
    const uint8_t* mask = masks + (size_t)rec.mask_id * ppad + (size_t)rec.row_min * detail_size + rec.col_min;
    if (vr > 0 && vc > 0) {
        const bool same = vr == bh && vc == bw;
        const bool whole = !same && vr % bh == 0 && vc % bw == 0 && vr / bh == vc / bw;
        const int f = whole ? vr / bh : 1;
        for (int p = threadIdx.x; p < bh * bw; p += kBlock) {
            const int r = p / bw, c = p - r * bw;
            if (mask[(size_t)r * detail_size + c] == 0) continue;          // (the value is needed under the mask only)
            int px[3];
            if (same || whole) {
                int sum[3] = {0, 0, 0};
                for (int i = 0; i < f; ++i)
                    for (int j = 0; j < f; ++j) {
                        const uint8_t* q = image + ((size_t)(y0 + r * f + i) * cols + (x0 + c * f + j)) * 3;
                        sum[0] += q[0]; sum[1] += q[1]; sum[2] += q[2];
                    }
                const float scale = 1.0f / (float)(f * f);
                for (int k = 0; k < 3; ++k)
                    px[k] = f == 1 ? sum[k] : (f == 2 ? (sum[k] + 2) >> 2 : min(255, max(0, __float2int_rn((float)sum[k] * scale))));
            } else {
                const AreaSpan ys = area_span(vr, bh, r), xs = area_span(vc, bw, c);
                float t[3] = {0.f, 0.f, 0.f};
                for (int ky = 0; ky < ys.count; ++ky) {
                    const uint8_t* row = image + ((size_t)(y0 + ys.first + ky) * cols + x0) * 3;
                    const float beta = ys.weight(ky);
                    float b[3] = {0.f, 0.f, 0.f};
                    for (int kx = 0; kx < xs.count; ++kx) {
                        const uint8_t* q = row + (size_t)(xs.first + kx) * 3;
                        const float al = xs.weight(kx);
                        b[0] = b[0] + (float)q[0] * al; b[1] = b[1] + (float)q[1] * al; b[2] = b[2] + (float)q[2] * al;
                    }
                    if (ky == 0) { t[0] = beta * b[0]; t[1] = beta * b[1]; t[2] = beta * b[2]; }
                    else { t[0] = t[0] + beta * b[0]; t[1] = t[1] + beta * b[1]; t[2] = t[2] + beta * b[2]; }
                }
                for (int k = 0; k < 3; ++k) px[k] = min(255, max(0, __float2int_rn(t[k])));
            }
            const int gray = (px[0] * 1868 + px[1] * 9617 + px[2] * 4899 + (1 << 13)) >> 14;
            atomicAdd(&hist[gray & 255], 1u);
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long count = 0;
        for (int i = 0; i < 256; ++i) count += hist[i];
        double entropy = 0.0;
        for (int i = 0; i < 256; ++i) {
            const double prob = (double)hist[i] / (double)count;        // count == 0: NaN, "> 0" is false, entropy stays 0
            if (prob > 0) entropy -= prob * log2(prob);
        }
        exceeded[cell] = entropy >= 8.0 * 0.7 ? 1 : 0;                  // MAX_ENTROPY * 0.7 (ImageUtility.h:69)
        if (entropy_out) entropy_out[cell] = entropy;
    }
}

// ---------------------------------------------------------------------------
// calibration: what a plain coalesced read stream reaches on this device
// ---------------------------------------------------------------------------

// Every lane sums float4 loads of a grid-stride stream (1 KiB per wave-instruction, the
// access pattern of the filter kernels' plane loads); one store per thread keeps the loads.
__global__ void __launch_bounds__(kBlock)
stream_read_kernel(const float4* __restrict__ src, size_t n_vec, float* __restrict__ sink)
{
    float4 acc{0.f, 0.f, 0.f, 0.f};
    const size_t stride = (size_t)gridDim.x * kBlock;
    size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x;
    for (; i + 3 * stride < n_vec; i += 4 * stride) {            // four loads in flight per lane
        const float4 a = src[i], b = src[i + stride], c = src[i + 2 * stride], d = src[i + 3 * stride];
        acc.x += (a.x + b.x) + (c.x + d.x); acc.y += (a.y + b.y) + (c.y + d.y);
        acc.z += (a.z + b.z) + (c.z + d.z); acc.w += (a.w + b.w) + (c.w + d.w);
    }
    for (; i < n_vec; i += stride) { const float4 a = src[i]; acc.x += a.x; acc.y += a.y; acc.z += a.z; acc.w += a.w; }
    sink[(size_t)blockIdx.x * kBlock + threadIdx.x] = (acc.x + acc.y) + (acc.z + acc.w);
}

// The box the fp32 filter's error band was validated on (engine.hip filter_band): f32 HWC pixels
// whose channel 0 lies in [-0.5, 100.5] and whose chroma sqrt(c1^2 + c2^2) is at most 150 (CIEDE2000 on Lab: the domain
// of the structured sweep and of the adversarial maximiser), or whose three channels lie within [-256, 256] (the
// Euclidean forms: three roundings of a value <= 442).  Bits OR-ed into *flags: 1 a non-finite value, 2 outside the
// CIEDE2000 box, 4 outside the Euclidean box.  The engine runs such inputs fp64-only (mm_timing.filter_off_range).
__global__ void __launch_bounds__(kBlock)
value_range_kernel(const float* __restrict__ hwc, size_t n_pixels, unsigned int* __restrict__ flags)
{
    unsigned int f = 0u;
    for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n_pixels; i += (size_t)gridDim.x * kBlock) {
        const float c0 = hwc[3 * i], c1 = hwc[3 * i + 1], c2 = hwc[3 * i + 2];
        if (!(fabsf(c0) <= 3.0e38f && fabsf(c1) <= 3.0e38f && fabsf(c2) <= 3.0e38f)) f |= 1u;       // NaN or infinite
        if (!(c0 >= -0.5f && c0 <= 100.5f && c1 * c1 + c2 * c2 <= 150.0f * 150.0f)) f |= 2u;
        if (!(fabsf(c0) <= 256.0f && fabsf(c1) <= 256.0f && fabsf(c2) <= 256.0f)) f |= 4u;
    }
    for (int off = 32; off > 0; off >>= 1) f |= (unsigned int)__shfl_xor((int)f, off, kWave);
    if ((threadIdx.x & 63) == 0 && f != 0u) atomicOr(flags, f);
}

// ---------------------------------------------------------------------------
// per-pair batch (KATs, differential tests)
// ---------------------------------------------------------------------------

// PARALLEL_FORM: what the parallel-order fp64 sums add per pair (colour_diff.hpp parallel_term: the algebraic form of CIEDE2000)
// instead of the reference form -- how tests measure one against the other (mm_set_option "probe_prepared" 3).
template <int TYPE, bool PARALLEL_FORM = false>
__global__ void __launch_bounds__(kBlock)
pair_diff_kernel(const float* __restrict__ first, const float* __restrict__ second,
                 size_t n, double* __restrict__ out)
{
    for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n;
         i += (size_t)gridDim.x * kBlock) {
        const float* p = first + 3 * i;
        const float* q = second + 3 * i;
        out[i] = PARALLEL_FORM ? parallel_term<TYPE>(Px{p[0], p[1], p[2]}, Px{q[0], q[1], q[2]})
                               : colour_diff<TYPE>(Px{p[0], p[1], p[2]}, Px{q[0], q[1], q[2]});
    }
}

}  // namespace mm
