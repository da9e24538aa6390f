// Calibration microbenchmark (diagnostics, not part of the product): issue rate of
// independent fp32 VALU instructions on gfx950 at a given occupancy.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/valu_rate tests/tools_valu_rate.hip && /tmp/valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>

template <int KIND>
__global__ void __launch_bounds__(256) rate_kernel(float* out, int iters, float a, float b)
{
    float x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
    for (int i = 0; i < iters; ++i) {
        if (KIND == 0) {          // fma
            x0 = fmaf(x0, a, b); x1 = fmaf(x1, a, b); x2 = fmaf(x2, a, b); x3 = fmaf(x3, a, b);
            x4 = fmaf(x4, a, b); x5 = fmaf(x5, a, b); x6 = fmaf(x6, a, b); x7 = fmaf(x7, a, b);
        } else if (KIND == 1) {   // rsq
            x0 = __builtin_amdgcn_rsqf(x0 + a); x1 = __builtin_amdgcn_rsqf(x1 + a); x2 = __builtin_amdgcn_rsqf(x2 + a); x3 = __builtin_amdgcn_rsqf(x3 + a);
            x4 = __builtin_amdgcn_rsqf(x4 + a); x5 = __builtin_amdgcn_rsqf(x5 + a); x6 = __builtin_amdgcn_rsqf(x6 + a); x7 = __builtin_amdgcn_rsqf(x7 + a);
        } else {                  // select (v_cndmask) + compare
            x0 = x0 > a ? x0 * b : x0 + b; x1 = x1 > a ? x1 * b : x1 + b; x2 = x2 > a ? x2 * b : x2 + b; x3 = x3 > a ? x3 * b : x3 + b;
            x4 = x4 > a ? x4 * b : x4 + b; x5 = x5 > a ? x5 * b : x5 + b; x6 = x6 > a ? x6 * b : x6 + b; x7 = x7 > a ? x7 * b : x7 + b;
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
}

template <int KIND>
void run(const char* name, int blocks_per_cu, int per_iter_instrs)
{
    const int cus = 256, iters = 20000;
    float* out;
    hipMalloc(&out, (size_t)cus * blocks_per_cu * 256 * sizeof(float));
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    rate_kernel<KIND><<<cus * blocks_per_cu, 256>>>(out, 100, 1.0001f, 0.5f);
    hipEventRecord(a);
    rate_kernel<KIND><<<cus * blocks_per_cu, 256>>>(out, iters, 1.0001f, 0.5f);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    // per SIMD: blocks_per_cu waves (256 threads = 4 waves = 1 per SIMD)
    const double wave_instrs_per_simd = (double)blocks_per_cu * iters * per_iter_instrs;
    printf("%-10s waves/SIMD=%d  %.2f ms  -> %.2f cycles per wave-instruction per SIMD (at 2.4 GHz)\n", name,
           blocks_per_cu, ms, ms * 1e-3 * 2.4e9 / wave_instrs_per_simd);
    hipFree(out);
}

int main()
{
    for (int occ : {1, 2, 4, 8}) {
        run<0>("fma", occ, 8);
        run<1>("add+rsq", occ, 16);
        run<2>("cmp+sel", occ, 32);
    }
    return 0;
}
