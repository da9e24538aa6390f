// Experiment (round 4; not part of the product): would a PACKED-fp16 pre-tier of the CIEDE2000 filter form
// pay?  The pre-tier would only take the pixel loop's early-exit decisions (with its own, wider band); what decides whether
// it is worth building is how much faster the per-pair arithmetic gets when two pixels share a lane-instruction.
//
// Two kernels in one harness, both at five waves per SIMD, both reading prepared operands from an L2-resident array (as
// the filter kernel reads a cell) and the hue table from LDS:
//   f32: mm::ciede2000_fast<true> exactly as filter_kernel runs it (colour_diff_fast.hpp), four pixels per iteration;
//   f16: the same form restated on half2 operands -- a, b scaled by 1/4 and the 7th powers by 1/256 to stay inside fp16's
//        range; the table index (11 bits + fraction) and the table interpolation stay in fp32 per pixel; the ten
//        transcendentals per pixel are v_*_f16 (one element per instruction, SDWA for the high half).
// Prints ns per pixel pair for both, their ratio, and the fp16 form's error against the fp32 form on the same pairs.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=fast -fno-slp-vectorize -Imosaicmagnifique_amd/csrc -o /tmp/fp16_form tools/experiments/r4_fp16_form.hip
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "colour_diff_fp16.hpp"

typedef _Float16 h1;
typedef _Float16 h2 __attribute__((ext_vector_type(2)));

namespace {

__device__ __forceinline__ h2 splat(float v) { return h2{(h1)v, (h1)v}; }
__device__ __forceinline__ unsigned int bits(h2 v) { return __builtin_bit_cast(unsigned int, v); }
__device__ __forceinline__ h2 from_bits(unsigned int v) { return __builtin_bit_cast(h2, v); }
__device__ __forceinline__ h2 hrsq(h2 v) { return h2{(h1)__builtin_amdgcn_rsqh(v.x), (h1)__builtin_amdgcn_rsqh(v.y)}; }
__device__ __forceinline__ h2 hrcp(h2 v) { return h2{(h1)__builtin_amdgcn_rcph(v.x), (h1)__builtin_amdgcn_rcph(v.y)}; }
__device__ __forceinline__ h2 hsqrt(h2 v) { return h2{(h1)__builtin_amdgcn_sqrth(v.x), (h1)__builtin_amdgcn_sqrth(v.y)}; }
// per-half select by a mask holding 0xffff / 0x0000 per half
__device__ __forceinline__ h2 sel(unsigned int mask, h2 a, h2 b) { return from_bits((bits(a) & mask) | (bits(b) & ~mask)); }

// sqrt(z / (z + 1)), z = s^7, s clamped to 4.8 (the ratio is 1 - 5e-6 there) and z carried as z / 256
__device__ __forceinline__ h2 ratio7(h2 s)
{
    s = __builtin_elementwise_min(s, splat(4.8f));
    const h2 s2 = s * s, s4 = s2 * s2;
    const h2 z = (s4 * splat(1.0f / 256.0f)) * s2 * s + splat(6e-8f);
    return z * hrsq(z * z + z * splat(1.0f / 256.0f));
}

// Two pixel pairs at once.  Operands prepared: l = L / 2 - 25, a = a / 4, b = b / 4, c = C / 50.
__device__ __forceinline__ h2 ciede2000_h2(h2 pl, h2 pa, h2 pb, h2 pc, h2 ql, h2 qa, h2 qb, h2 qc, const mm::HueEntry* __restrict__ lut)
{
    const h2 one_g = splat(1.5f) - splat(0.5f) * ratio7(pc + qc);
    const h2 a1 = one_g * pa, a2 = one_g * qa;
    const h2 m1 = a1 * a1 + (pb * pb + splat(6e-8f)), m2 = a2 * a2 + (qb * qb + splat(6e-8f));
    const h2 c1p = hsqrt(m1), c2p = hsqrt(m2);
    const h2 dot = a1 * a2 + pb * qb;
    const h2 cross = a1 * qb - pb * a2;
    // per-half masks from the sign bits: 0xffff where dot < 0 (obtuse)
    const unsigned int ob = (bits(dot) & 0x80008000u) >> 15;
    const unsigned int obtuse = (ob << 16) - ob;
    const unsigned int sgn = bits(cross) & 0x80008000u;
    const h2 e = c1p * c2p + from_bits(bits(dot) & 0x7fff7fffu);
    const h2 r = hrsq(splat(0.5f) * e);
    const h2 d_h = sel(obtuse, from_bits(bits(e) | sgn), cross) * r;
    const h2 c2s = from_bits(bits(c2p) ^ (obtuse & 0x80008000u));
    const h2 wx = a2 * c1p + (a1 * c2s + splat(6e-8f)), wy = qb * c1p + pb * c2s;
    const h2 hx = sel(obtuse, from_bits(bits(wy) ^ sgn), wx);
    const h2 hy = sel(obtuse, from_bits(bits(wx) ^ sgn ^ 0x80008000u), wy);
    // table index and interpolation per pixel in fp32 (the index needs 11 bits and a fraction)
    float tt[2], ss[2];
    for (int k = 0; k < 2; ++k) {
        const float fx = (float)(k ? hx.y : hx.x), fy = (float)(k ? hy.y : hy.x);
        const float qd = fx * mm::frcp(fabsf(fx) + fabsf(fy));
        const float x = fmaf(-((float)(mm::kHueLutSize / 4) - 0x1p-13f), copysignf(1.0f + qd, fy), (float)(mm::kHueLutSize / 2));
        const int idx = (int)x;
        const float fr = __builtin_amdgcn_fractf(x);
        const mm::HueEntry e0 = lut[idx], e1 = lut[idx + 1];
        tt[k] = fmaf(fr, e1.x - e0.x, e0.x);
        ss[k] = fmaf(fr, e1.y - e0.y, e0.y);
    }
    const h2 t = h2{(h1)tt[0], (h1)tt[1]}, s2 = h2{(h1)ss[0], (h1)ss[1]};
    const h2 csum = c1p + c2p;                              // (C1' + C2') / 4
    const h2 r_t = s2 * ratio7(csum * splat(0.08f));
    const h2 lb = pl + ql;
    const h2 u_l = lb * lb + splat(20.0f);
    const h2 s_l = (splat(0.0075f) * u_l - splat(0.15f)) * hrsq(u_l) + splat(0.5f);
    const h2 s_c = splat(0.09f) * csum + splat(1.0f);
    const h2 s_h = (splat(4.0f) * csum) * t + splat(1.0f);
    const h2 d_l = ql - pl, d_c = c2p - c1p;
    const h2 big_d = s_c * s_h, ta = d_c * s_h, tb = d_h * s_c;
    const h2 num = ta * (r_t * tb + ta) + tb * tb;
    const h2 inv_d = hrcp(big_d) * splat(4.0f);              // the chroma / hue differences were carried at 1 / 4
    const h2 t_l = d_l * hrcp(s_l);
    const h2 xx = t_l * t_l + num * (inv_d * inv_d);
    return hsqrt(__builtin_elementwise_max(xx, splat(0.0f)));
}

constexpr int kThreads = 256;

// operands: [n][8][4] floats -- per float4 slot: cell L', a, b, C/50 and library L', a, b, C/50 (prepared as interleave_kernel does)
__global__ void __launch_bounds__(kThreads) __attribute__((amdgpu_waves_per_eu(5, 5)))
form_f32(const float4* __restrict__ ops, int n_slots, int iters, const mm::HueEntry* __restrict__ hue_lut, float* __restrict__ out)
{
    __shared__ mm::HueEntry lut[mm::kHueLutEntries];
    for (int i = threadIdx.x; i < mm::kHueLutEntries; i += kThreads) lut[i] = hue_lut[i];
    __syncthreads();
    float acc = 0.0f;
    for (int it = 0; it < iters; ++it) {
        const float4* o = ops + (size_t)(((it * 977 + blockIdx.x) % n_slots) * kThreads + threadIdx.x) * 8;
        const float4 p0 = o[0], p1 = o[1], p2 = o[2], pc = o[3], q0 = o[4], q1 = o[5], q2 = o[6], qc = o[7];
        bool e0, e1, e2, e3;
        const float v0 = mm::ciede2000_fast<true>(mm::PxF{p0.x, p1.x, p2.x}, mm::PxF{q0.x, q1.x, q2.x}, pc.x, qc.x, lut, &e0);
        const float v1 = mm::ciede2000_fast<true>(mm::PxF{p0.y, p1.y, p2.y}, mm::PxF{q0.y, q1.y, q2.y}, pc.y, qc.y, lut, &e1);
        const float v2 = mm::ciede2000_fast<true>(mm::PxF{p0.z, p1.z, p2.z}, mm::PxF{q0.z, q1.z, q2.z}, pc.z, qc.z, lut, &e2);
        const float v3 = mm::ciede2000_fast<true>(mm::PxF{p0.w, p1.w, p2.w}, mm::PxF{q0.w, q1.w, q2.w}, pc.w, qc.w, lut, &e3);
        acc += (v0 + v1) + (v2 + v3);
        if (__builtin_amdgcn_ballot_w64(e0 || e1 || e2 || e3) != 0) acc += 1e-30f;      // (the kernel's wave-wide test)
    }
    out[(size_t)blockIdx.x * kThreads + threadIdx.x] = acc;
}

// operands as halfs: [n][8] x half4 (two half2) per lane: 64 B per lane and iteration instead of 128
__global__ void __launch_bounds__(kThreads) __attribute__((amdgpu_waves_per_eu(5, 5)))
form_f16(const uint2* __restrict__ ops, int n_slots, int iters, const mm::HueEntry* __restrict__ hue_lut, float* __restrict__ out)
{
    __shared__ mm::HueEntry lut[mm::kHueLutEntries];
    for (int i = threadIdx.x; i < mm::kHueLutEntries; i += kThreads) lut[i] = hue_lut[i];
    __syncthreads();
    float acc = 0.0f;
    for (int it = 0; it < iters; ++it) {
        const uint2* o = ops + (size_t)(((it * 977 + blockIdx.x) % n_slots) * kThreads + threadIdx.x) * 8;
        const uint2 p0 = o[0], p1 = o[1], p2 = o[2], pc = o[3], q0 = o[4], q1 = o[5], q2 = o[6], qc = o[7];
        const h2 va = ciede2000_h2(from_bits(p0.x), from_bits(p1.x), from_bits(p2.x), from_bits(pc.x),
                                   from_bits(q0.x), from_bits(q1.x), from_bits(q2.x), from_bits(qc.x), lut);
        const h2 vb = ciede2000_h2(from_bits(p0.y), from_bits(p1.y), from_bits(p2.y), from_bits(pc.y),
                                   from_bits(q0.y), from_bits(q1.y), from_bits(q2.y), from_bits(qc.y), lut);
        const h2 s = va + vb;                       // one packed add, then two conversions: the group's sum goes on in fp32
        acc += (float)s.x + (float)s.y;
    }
    out[(size_t)blockIdx.x * kThreads + threadIdx.x] = acc;
}

// error of the fp16 form against the fp32 form, pair by pair
__global__ void __launch_bounds__(kThreads)
form_error(const float4* __restrict__ ops32, const uint2* __restrict__ ops16, int n, const mm::HueEntry* __restrict__ hue_lut,
           float* __restrict__ err_out, float* __restrict__ val_out)
{
    __shared__ mm::HueEntry lut[mm::kHueLutEntries];
    for (int i = threadIdx.x; i < mm::kHueLutEntries; i += kThreads) lut[i] = hue_lut[i];
    __syncthreads();
    const int t = blockIdx.x * kThreads + threadIdx.x;
    if (t >= n) return;
    const float4* o = ops32 + (size_t)t * 8;
    const uint2* h = ops16 + (size_t)t * 8;
    bool e;
    const float v0 = mm::ciede2000_fast<true>(mm::PxF{o[0].x, o[1].x, o[2].x}, mm::PxF{o[4].x, o[5].x, o[6].x}, o[3].x, o[7].x, lut, &e);
    const h2 w = ciede2000_h2(from_bits(h[0].x), from_bits(h[1].x), from_bits(h[2].x), from_bits(h[3].x),
                              from_bits(h[4].x), from_bits(h[5].x), from_bits(h[6].x), from_bits(h[7].x), lut);
    err_out[t] = e ? 0.0f : fabsf((float)w.x - v0);
    val_out[t] = v0;
}

std::vector<float> build_hue_lut()
{
    constexpr int N = mm::kHueLutSize;
    constexpr double pi = 3.14159265358979323846;
    auto rad = [&](double d) { return d * pi / 180.0; };
    std::vector<float> lut((size_t)mm::kHueLutEntries * 2);
    for (int i = 0; i < mm::kHueLutEntries; ++i) {
        const double t = 4.0 * (i % N) / N;
        const bool upper = t <= 2.0;
        const double q = upper ? 1.0 - t : t - 3.0;
        const double hc = q, hs = (upper ? 1.0 : -1.0) * (1.0 - std::fabs(q));
        double h = std::atan2(hs, hc);
        if (h < 0.0) h += 2.0 * pi;
        const double T = 1.0 - 0.17 * std::cos(h - rad(30.0)) + 0.24 * std::cos(2.0 * h) + 0.32 * std::cos(3.0 * h + rad(6.0)) - 0.20 * std::cos(4.0 * h - rad(63.0));
        const double z = (h - rad(275.0)) / rad(25.0);
        lut[(size_t)2 * i] = (float)(0.0075 * T);
        lut[(size_t)2 * i + 1] = (float)(-2.0 * std::sin(2.0 * rad(30.0) * std::exp(-z * z)));
    }
    return lut;
}

unsigned short to_half(float f)
{
    const _Float16 h = (_Float16)f;
    unsigned short u;
    __builtin_memcpy(&u, &h, 2);
    return u;
}

}  // namespace

int main()
{
    const int n_slots = 64, lanes = n_slots * kThreads;      // 64 x 256 lanes x 128 B = 2 MB of operands: L2-resident
    std::vector<float> ops32((size_t)lanes * 32);
    std::vector<unsigned short> ops16((size_t)lanes * 32);
    srand(7);
    auto uni = [](float lo, float hi) { return lo + (hi - lo) * (float)rand() / (float)RAND_MAX; };
    for (int t = 0; t < lanes; ++t)
        for (int px = 0; px < 4; ++px) {
            // synthetic-benchmark-like pairs: a cell pixel and a library pixel some way apart
            const float L1 = uni(5, 95), a1 = uni(-80, 80), b1 = uni(-80, 80);
            const float L2 = L1 + uni(-30, 30), a2 = a1 + uni(-40, 40), b2 = b1 + uni(-40, 40);
            const float v[8] = {0.5f * L1 - 25.f, a1, b1, 0.02f * std::sqrt(a1 * a1 + b1 * b1),
                                0.5f * L2 - 25.f, a2, b2, 0.02f * std::sqrt(a2 * a2 + b2 * b2)};
            for (int k = 0; k < 8; ++k) {
                ops32[((size_t)t * 8 + k) * 4 + px] = v[k];
                const bool ab = (k & 3) == 1 || (k & 3) == 2;
                ops16[((size_t)t * 8 + k) * 4 + px] = to_half(ab ? 0.25f * v[k] : v[k]);
            }
        }
    const std::vector<float> lut = build_hue_lut();
    float4* d32; uint2* d16; mm::HueEntry* dlut; float *dout, *derr, *dval;
    (void)hipMalloc(&d32, ops32.size() * 4); (void)hipMalloc(&d16, ops16.size() * 2); (void)hipMalloc(&dlut, lut.size() * 4);
    const int blocks = 256 * 5;
    (void)hipMalloc(&dout, (size_t)blocks * kThreads * 4); (void)hipMalloc(&derr, (size_t)lanes * 4); (void)hipMalloc(&dval, (size_t)lanes * 4);
    (void)hipMemcpy(d32, ops32.data(), ops32.size() * 4, hipMemcpyHostToDevice);
    (void)hipMemcpy(d16, ops16.data(), ops16.size() * 2, hipMemcpyHostToDevice);
    (void)hipMemcpy(dlut, lut.data(), lut.size() * 4, hipMemcpyHostToDevice);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int iters = 4000;
    double ns[2] = {0, 0};
    for (int rep = 0; rep < 3; ++rep)
        for (int which = 0; which < 2; ++which) {
            if (which == 0) form_f32<<<blocks, kThreads>>>(d32, n_slots, 50, dlut, dout); else form_f16<<<blocks, kThreads>>>(d16, n_slots, 50, dlut, dout);
            (void)hipEventRecord(e0);
            if (which == 0) form_f32<<<blocks, kThreads>>>(d32, n_slots, iters, dlut, dout); else form_f16<<<blocks, kThreads>>>(d16, n_slots, iters, dlut, dout);
            (void)hipEventRecord(e1);
            (void)hipEventSynchronize(e1);
            float ms; (void)hipEventElapsedTime(&ms, e0, e1);
            const double pairs = (double)blocks * kThreads * iters * 4;
            ns[which] = ms * 1e6 / pairs;
            printf("%s: %.3f ms for %.3g pixel pairs = %.5f ns per pair (whole device), %.3g pairs/s\n", which ? "f16 packed" : "f32       ", ms, pairs, ns[which], pairs / (ms * 1e-3));
        }
    printf("ratio f32 / f16 = %.3f\n", ns[0] / ns[1]);
    form_error<<<(lanes + kThreads - 1) / kThreads, kThreads>>>(d32, d16, lanes, dlut, derr, dval);
    std::vector<float> err(lanes), val(lanes);
    (void)hipMemcpy(err.data(), derr, (size_t)lanes * 4, hipMemcpyDeviceToHost);
    (void)hipMemcpy(val.data(), dval, (size_t)lanes * 4, hipMemcpyDeviceToHost);
    double worst = 0, sum = 0, worst_rel = 0, signed_sum = 0;
    for (int t = 0; t < lanes; ++t) {
        if (!(err[t] == err[t])) { printf("NaN at %d\n", t); continue; }
        worst = std::max(worst, (double)err[t]); sum += err[t];
        worst_rel = std::max(worst_rel, (double)err[t] / std::max(1.0, (double)val[t]));
    }
    (void)signed_sum;
    printf("fp16 form vs fp32 form on %d pairs: max |err| %.4f, mean |err| %.5f, max |err| / max(1, dE) %.5f\n", lanes, worst, sum / lanes, worst_rel);
    return 0;
}
