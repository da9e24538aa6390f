#include <hip/hip_runtime.h>
#include <cstdio>
#include "colour_diff_fp16.hpp"
using namespace mm;
__global__ void k(const unsigned int* in, unsigned int* out, int n)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    const half2_t v = h2_from(in[t]);
    const half2_t a = h2_rsq(v), b = h2_rcp(v), c = h2_sqrt(v);
    const half2_t ra{(half1_t)__builtin_amdgcn_rsqh(v.x), (half1_t)__builtin_amdgcn_rsqh(v.y)};
    const half2_t rb{(half1_t)__builtin_amdgcn_rcph(v.x), (half1_t)__builtin_amdgcn_rcph(v.y)};
    const half2_t rc{(half1_t)__builtin_amdgcn_sqrth(v.x), (half1_t)__builtin_amdgcn_sqrth(v.y)};
    // chained use right after, as the form does
    const half2_t d = h2_rsq(v * v + v) * v;
    const half2_t vv = v * v + v;
    const half2_t rd = half2_t{(half1_t)__builtin_amdgcn_rsqh(vv.x), (half1_t)__builtin_amdgcn_rsqh(vv.y)} * v;
    out[8 * t + 0] = h2_bits(a); out[8 * t + 1] = h2_bits(ra); out[8 * t + 2] = h2_bits(b); out[8 * t + 3] = h2_bits(rb);
    out[8 * t + 4] = h2_bits(c); out[8 * t + 5] = h2_bits(rc); out[8 * t + 6] = h2_bits(d); out[8 * t + 7] = h2_bits(rd);
}
int main()
{
    const int n = 1 << 16;
    unsigned int *din, *dout;
    hipMalloc(&din, n * 4); hipMalloc(&dout, n * 32);
    unsigned int* h = new unsigned int[n];
    for (int i = 0; i < n; ++i) { _Float16 lo = (_Float16)(0.01f + i * 0.003f), hi = (_Float16)(200.0f - i * 0.002f); unsigned short a, b; __builtin_memcpy(&a, &lo, 2); __builtin_memcpy(&b, &hi, 2); h[i] = a | ((unsigned)b << 16); }
    hipMemcpy(din, h, n * 4, hipMemcpyHostToDevice);
    k<<<n / 256, 256>>>(din, dout, n);
    unsigned int* o = new unsigned int[n * 8];
    hipMemcpy(o, dout, n * 32, hipMemcpyDeviceToHost);
    int bad[4] = {0, 0, 0, 0};
    for (int i = 0; i < n; ++i) for (int q = 0; q < 4; ++q) if (o[8 * i + 2 * q] != o[8 * i + 2 * q + 1]) { if (bad[q] < 3) printf("mismatch op %d at %d: in %08x got %08x want %08x\n", q, i, h[i], o[8 * i + 2 * q], o[8 * i + 2 * q + 1]); ++bad[q]; }
    printf("mismatches rsq %d rcp %d sqrt %d chained %d of %d\n", bad[0], bad[1], bad[2], bad[3], n);
    return 0;
}
