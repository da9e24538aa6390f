// filter.hip -- tier 1 of the two-tier scoring: fp32 filter scores of every
// (cell, candidate).  Built WITH fp contraction (fused multiply-add is wanted
// here); the fp64 reference-order kernels live in engine.hip, built without.
//
// Nothing in this translation unit decides a best fit: see colour_diff_fast.hpp.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "colour_diff.hpp"
#include "colour_diff_fast.hpp"
#include "launch.hpp"

namespace mm {

namespace {

constexpr int kBlock = 256;
constexpr int kWave = 64;

__device__ __forceinline__ float wave_sum_f(float v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, kWave);
    return v;
}

// One pair through the fp32 form.  For CIEDE2000 `edge` reports a pair on the
// opposite-hue discontinuity (colour_diff_fast.hpp): the filter kernel logs it and
// edge_fix_kernel later replaces its fp32 term by the fp64 one.
template <int TYPE>
__device__ __forceinline__ float pair_fast(const PxF& p, const PxF& q, bool* edge)
{
    if constexpr (TYPE == CIEDE2000) {
        return ciede2000_fast(p, q, edge);
    } else {
        *edge = false;
        return euclidean_fast(p, q);
    }
}

// Same work decomposition as score_kernel (kernels.hpp): one wave per
// (cell, candidate), lanes over float4 pixel groups, 4 waves = 4 candidates of
// one cell; cells fastest within a group so resident workgroups share library tiles.
template <int TYPE>
__global__ void __launch_bounds__(kBlock)
filter_kernel(const float* __restrict__ cells, const uint8_t* __restrict__ cmask,
              const float* __restrict__ lib, float* __restrict__ scores,
              int n_cells, int L, int ppad, int cand_per_block, int cell_group,
              EdgeEntry* __restrict__ edges, unsigned int* __restrict__ edge_count,
              unsigned int edge_cap)
{
    const int n_tiles = (L + cand_per_block - 1) / cand_per_block;
    const long long bid = blockIdx.x;
    const long long per_group = (long long)cell_group * n_tiles;
    const int group = (int)(bid / per_group);
    const long long rem = bid - (long long)group * per_group;
    const int cells_here = min(cell_group, n_cells - group * cell_group);
    const int tile = (int)(rem / cells_here);
    const int cell = group * cell_group + (int)(rem - (long long)tile * cells_here);
    if (tile >= n_tiles || cell >= n_cells) return;

    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int groups = ppad / (4 * kWave);
    const float4* c0 = reinterpret_cast<const float4*>(cells + (size_t)cell * 3 * ppad);
    const float4* c1 = c0 + ppad / 4;
    const float4* c2 = c1 + ppad / 4;
    const uint32_t* mk = reinterpret_cast<const uint32_t*>(cmask + (size_t)cell * ppad);

    const int l_end = min((tile + 1) * cand_per_block, L);
    for (int l = tile * cand_per_block + wave; l < l_end; l += kBlock / kWave) {
        const float4* b0 = reinterpret_cast<const float4*>(lib + (size_t)l * 3 * ppad);
        const float4* b1 = b0 + ppad / 4;
        const float4* b2 = b1 + ppad / 4;
        float acc = 0.0f;
        for (int j = 0; j < groups; ++j) {
            const int q = lane + kWave * j;
            const uint32_t m = mk[q];
            if (m == 0) continue;
            const float4 p0 = c0[q], p1 = c1[q], p2 = c2[q];
            const float4 q0 = b0[q], q1 = b1[q], q2 = b2[q];
            bool e0, e1, e2, e3;
            const float v0 = pair_fast<TYPE>(PxF{p0.x, p1.x, p2.x}, PxF{q0.x, q1.x, q2.x}, &e0);
            const float v1 = pair_fast<TYPE>(PxF{p0.y, p1.y, p2.y}, PxF{q0.y, q1.y, q2.y}, &e1);
            const float v2 = pair_fast<TYPE>(PxF{p0.z, p1.z, p2.z}, PxF{q0.z, q1.z, q2.z}, &e2);
            const float v3 = pair_fast<TYPE>(PxF{p0.w, p1.w, p2.w}, PxF{q0.w, q1.w, q2.w}, &e3);
            acc += ((m & 0x000000ffu) ? v0 : 0.0f) + ((m & 0x0000ff00u) ? v1 : 0.0f)
                 + ((m & 0x00ff0000u) ? v2 : 0.0f) + ((m & 0xff000000u) ? v3 : 0.0f);
            if constexpr (TYPE == CIEDE2000) {
                const uint32_t em = (e0 ? 0x000000ffu : 0u) | (e1 ? 0x0000ff00u : 0u) |
                                   (e2 ? 0x00ff0000u : 0u) | (e3 ? 0xff000000u : 0u);
                if (em & m) {                               // rare: ~1e-5 of pairs
                    for (int k = 0; k < 4; ++k)
                        if ((em & m) >> (8 * k) & 0xffu) {
                            const unsigned int slot = atomicAdd(edge_count, 1u);
                            if (slot < edge_cap) edges[slot] = EdgeEntry{cell, l, 4 * q + k};
                        }
                }
            }
        }
        acc = wave_sum_f(acc);
        if (lane == 0) scores[(size_t)cell * L + l] = acc;
    }
}

// Replaces the fp32 term of every logged discontinuity pair by the fp64
// reference-form value: scores[cell][cand] += (float)(exact - fast).
__global__ void __launch_bounds__(kBlock)
edge_fix_kernel(const float* __restrict__ cells, const float* __restrict__ lib,
                float* __restrict__ scores, int L, int ppad,
                const EdgeEntry* __restrict__ edges, const unsigned int* __restrict__ edge_count,
                unsigned int edge_cap)
{
    const unsigned int n = min(*edge_count, edge_cap);
    for (unsigned int i = blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock) {
        const EdgeEntry en = edges[i];
        const float* c = cells + (size_t)en.cell * 3 * ppad + en.pixel;
        const float* b = lib + (size_t)en.cand * 3 * ppad + en.pixel;
        const PxF p{c[0], c[ppad], c[2 * (size_t)ppad]}, q{b[0], b[ppad], b[2 * (size_t)ppad]};
        bool edge;
        const float fast = ciede2000_fast(p, q, &edge);
        const double exact = ciede2000(Px{p.c0, p.c1, p.c2}, Px{q.c0, q.c1, q.c2});
        atomicAdd(&scores[(size_t)en.cell * L + en.cand], (float)(exact - (double)fast));
    }
}

template <int TYPE>
__global__ void __launch_bounds__(kBlock)
pair_fast_kernel(const float* __restrict__ first, const float* __restrict__ second,
                 size_t n, float* __restrict__ out, int flag_edges)
{
    for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n;
         i += (size_t)gridDim.x * kBlock) {
        const float* p = first + 3 * i;
        const float* q = second + 3 * i;
        bool edge;
        float v = pair_fast<TYPE>(PxF{p[0], p[1], p[2]}, PxF{q[0], q[1], q[2]}, &edge);
        if (edge && flag_edges) v = -1.0f;      // diagnostics: show where the fp64 form takes over
        out[i] = v;
    }
}

}  // namespace

hipError_t launch_filter(int type, const float* cells, const uint8_t* cmask, const float* lib,
                         float* scores, int n_cells, int L, int ppad,
                         EdgeEntry* edges, unsigned int* edge_count, unsigned int edge_cap,
                         hipStream_t stream)
{
    const int cand_per_block = 16, cell_group = 512;
    const long long n_tiles = (L + cand_per_block - 1) / cand_per_block;
    const long long blocks = (long long)n_cells * n_tiles;
    if (blocks <= 0 || blocks > 0x7fffffffLL) return hipErrorInvalidValue;
    hipError_t code = hipMemsetAsync(edge_count, 0, sizeof(unsigned int), stream);
    if (code != hipSuccess) return code;
    if (type == CIEDE2000) {
        filter_kernel<CIEDE2000><<<dim3((unsigned)blocks), kBlock, 0, stream>>>(
            cells, cmask, lib, scores, n_cells, L, ppad, cand_per_block, cell_group, edges, edge_count, edge_cap);
        if ((code = hipGetLastError()) != hipSuccess) return code;
        edge_fix_kernel<<<1024, kBlock, 0, stream>>>(cells, lib, scores, L, ppad, edges, edge_count, edge_cap);
    } else {
        filter_kernel<RGB_EUCLIDEAN><<<dim3((unsigned)blocks), kBlock, 0, stream>>>(
            cells, cmask, lib, scores, n_cells, L, ppad, cand_per_block, cell_group, edges, edge_count, edge_cap);
    }
    return hipGetLastError();
}

hipError_t launch_pair_fast(int type, const float* first, const float* second, size_t n,
                            float* out, int flag_edges, hipStream_t stream)
{
    const int blocks = (int)((n + kBlock - 1) / kBlock < 8192 ? (n + kBlock - 1) / kBlock : 8192);
    if (type == CIEDE2000)
        pair_fast_kernel<CIEDE2000><<<blocks, kBlock, 0, stream>>>(first, second, n, out, flag_edges);
    else
        pair_fast_kernel<RGB_EUCLIDEAN><<<blocks, kBlock, 0, stream>>>(first, second, n, out, flag_edges);
    return hipGetLastError();
}

}  // namespace mm
