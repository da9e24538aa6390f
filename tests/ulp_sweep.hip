// ulp_sweep.hip -- test program (not part of the product library): how accurate are the hardware instructions that
// tools/filter_error_bound.py charges with "one ulp"?  Exhaustive where the operand space allows it:
//   v_rcp_f32, v_rsq_f32, v_sqrt_f32    every positive normal float (2^31 - 2^24 operands each; v_rcp where the result is normal too)
//   v_exp_f32 (2^x)                     every float in [-64, 0]  (stats_bound: exp2 of a non-positive exponent)
//   v_sin_f32 (sin(2 pi x))             every float in [2^-20, 1/6]  (stats_bound: x = E / 6, E in (0, 1]; absolute error)
//   v_rcp_f16, v_rsq_f16, v_sqrt_f16    every positive normal half (the packed-fp16 pre-tier)
// against the correctly rounded fp64 result.  Prints, per instruction, the largest error in units of the RESULT's last place
// (relative error x 2^23, resp. x 2^10, of a result normalised to [1, 2) is at most that many ulp) and the operand that gave it.
// Built and run by tests/test_gpu_transcendentals.py.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdint>
#include <cstring>

struct Worst { unsigned long long key; };      // (error in 1/1024 ulp) << 32 | operand bits

__device__ inline void note(unsigned long long* slot, double ulps, unsigned int bits)
{
    const unsigned long long e = (unsigned long long)fmin(ulps * 1024.0, 4.0e9);
    atomicMax(slot, (e << 32) | bits);
}

// error of `got` against `want` in ulps of want's binade (p = mantissa bits: 24 single, 11 half)
__device__ inline double ulps_of(double got, double want, int p)
{
    if (!(want == want) || want == 0.0) return 0.0;
    int ex;
    (void)frexp(want, &ex);                       // want = m 2^ex, m in [0.5, 1)
    return fabs(got - want) / ldexp(1.0, ex - p);
}

__global__ void sweep_f32(unsigned long long* worst)
{
    const unsigned long long n = 0x7f800000ull - 0x00800000ull;      // positive normal floats
    for (unsigned long long i = blockIdx.x * (unsigned long long)blockDim.x + threadIdx.x; i < n; i += (unsigned long long)gridDim.x * blockDim.x) {
        const unsigned int bits = 0x00800000u + (unsigned int)i;
        const float x = __uint_as_float(bits);
        const double xd = (double)x;
        if (x <= 0x1p126f) note(worst + 0, ulps_of((double)__builtin_amdgcn_rcpf(x), 1.0 / xd, 24), bits);      // (a NORMAL result: below 2^-126 the mode flushes)
        note(worst + 1, ulps_of((double)__builtin_amdgcn_rsqf(x), 1.0 / sqrt(xd), 24), bits);
        note(worst + 2, ulps_of((double)__builtin_amdgcn_sqrtf(x), sqrt(xd), 24), bits);
        if (x <= 64.0f) {                         // exp2 of -x, x in (0, 64]: results in [2^-64, 1)
            note(worst + 3, ulps_of((double)__builtin_amdgcn_exp2f(-x), exp2(-xd), 24), bits);
        }
        if (x >= 0x1p-20f && x <= (1.0f / 6.0f)) {     // sin(2 pi x): ABSOLUTE error in units of 2^-24 (what stats_bound's cushion is sized in)
            const double want = sin(6.283185307179586476925286766559 * xd);
            note(worst + 4, fabs((double)__builtin_amdgcn_sinf(x) - want) * 16777216.0, bits);
        }
    }
}

__global__ void sweep_f16(unsigned long long* worst)
{
    for (unsigned int bits = 0x0400u + blockIdx.x * blockDim.x + threadIdx.x; bits < 0x7c00u; bits += gridDim.x * blockDim.x) {     // positive normal halfs
        const _Float16 x = __builtin_bit_cast(_Float16, (unsigned short)bits);
        const double xd = (double)x;
        if (xd <= 16384.0) note(worst + 5, ulps_of((double)__builtin_amdgcn_rcph(x), 1.0 / xd, 11), bits);        // (a normal result, >= 2^-14: the pre-tier's reciprocal takes operands in [0.5, 64])
        note(worst + 6, ulps_of((double)__builtin_amdgcn_rsqh(x), 1.0 / sqrt(xd), 11), bits);
        note(worst + 7, ulps_of((double)__builtin_amdgcn_sqrth(x), sqrt(xd), 11), bits);
    }
}

int main()
{
    unsigned long long* worst = nullptr;
    if (hipMalloc(&worst, 8 * sizeof(unsigned long long)) != hipSuccess) { fprintf(stderr, "hipMalloc failed\n"); return 2; }
    (void)hipMemset(worst, 0, 8 * sizeof(unsigned long long));
    sweep_f32<<<4096, 256>>>(worst);
    sweep_f16<<<64, 256>>>(worst);
    if (hipDeviceSynchronize() != hipSuccess) { fprintf(stderr, "kernel failed\n"); return 3; }
    unsigned long long h[8];
    (void)hipMemcpy(h, worst, sizeof h, hipMemcpyDeviceToHost);
    const char* names[8] = {"v_rcp_f32", "v_rsq_f32", "v_sqrt_f32", "v_exp_f32", "v_sin_f32_abs_2^-24", "v_rcp_f16", "v_rsq_f16", "v_sqrt_f16"};
    for (int i = 0; i < 8; ++i)
        printf("%s %.4f 0x%08x\n", names[i], (double)(h[i] >> 32) / 1024.0, (unsigned int)(h[i] & 0xffffffffu));
    return 0;
}
