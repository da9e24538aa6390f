// Calibration microbenchmark (diagnostics, not part of the product): issue cost of VALU
// instructions on gfx950, per wave-instruction per SIMD, at a given number of resident
// waves per SIMD.  Every kernel runs 8 independent instructions of one kind (or a mix)
// per loop iteration; time from s_memtime inside the kernel and from HIP events.
//   hipcc --offload-arch=gfx950 -O3 -w -o build/valu_rate tools/valu_rate.hip && build/valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

// "op dst, dst <rest>" for dst = %0..%7
#define I1(op, d, rest) op " %" #d ", %" #d rest "\n"
#define I8(op, rest) I1(op, 0, rest) I1(op, 1, rest) I1(op, 2, rest) I1(op, 3, rest) I1(op, 4, rest) I1(op, 5, rest) I1(op, 6, rest) I1(op, 7, rest)
// "op dst, <pre>, dst"
#define J1(op, d, pre) op " %" #d ", " pre ", %" #d "\n"
#define J8(op, pre) J1(op, 0, pre) J1(op, 1, pre) J1(op, 2, pre) J1(op, 3, pre) J1(op, 4, pre) J1(op, 5, pre) J1(op, 6, pre) J1(op, 7, pre)
// mixes: two different ops alternating (A on even, B on odd registers)
#define M8(opa, resta, opb, restb) I1(opa, 0, resta) I1(opb, 1, restb) I1(opa, 2, resta) I1(opb, 3, restb) I1(opa, 4, resta) I1(opb, 5, restb) I1(opa, 6, resta) I1(opb, 7, restb)
// 3 of A then 1 of B, twice
#define M31(opa, resta, opb, restb) I1(opa, 0, resta) I1(opa, 1, resta) I1(opa, 2, resta) I1(opb, 3, restb) I1(opa, 4, resta) I1(opa, 5, resta) I1(opa, 6, resta) I1(opb, 7, restb)

#define KERNEL(NAME, ASM)                                                                                     \
    __global__ void __launch_bounds__(1024) NAME(long long* cyc, float* out, int iters, float a, float b,     \
                                                 unsigned long long m, float sa)                              \
    {                                                                                                         \
        float x0 = threadIdx.x + 1.5f, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5,      \
              x6 = x0 + 6, x7 = x0 + 7;                                                                       \
        const long long t0 = __builtin_amdgcn_s_memtime();                                                    \
        for (int i = 0; i < iters; ++i)                                                                       \
            asm volatile(ASM : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) \
                         : "v"(a), "v"(b), "s"(m), "s"(sa) : "vcc", "s20", "s21", "s22", "s23");              \
        const long long t1 = __builtin_amdgcn_s_memtime();                                                    \
        out[blockIdx.x * blockDim.x + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;                   \
        if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;        \
    }

KERNEL(k_fma, I8("v_fma_f32", ", %8, %9"))
KERNEL(k_fma_mod, I8("v_fma_f32", ", |%8|, -%9"))
KERNEL(k_fma_sgpr, I8("v_fma_f32", ", %11, %9"))
KERNEL(k_fmac, J8("v_fmac_f32", "%8"))
KERNEL(k_mul, I8("v_mul_f32", ", %8"))
KERNEL(k_mul_lit, J8("v_mul_f32", "0x3d23d70a"))
KERNEL(k_mul_inl, J8("v_mul_f32", "0.5"))
KERNEL(k_add, I8("v_add_f32", ", %8"))
KERNEL(k_sub, I8("v_sub_f32", ", %8"))
KERNEL(k_add_abs, I8("v_add_f32", ", |%8|"))
KERNEL(k_max, I8("v_max_f32", ", %8"))
KERNEL(k_min, I8("v_min_f32", ", %8"))
KERNEL(k_max3, I8("v_max3_f32", ", %8, %9"))
KERNEL(k_med3, I8("v_med3_f32", ", %8, %9"))
KERNEL(k_cndmask_s, I8("v_cndmask_b32", ", %8, %10"))
KERNEL(k_cndmask_vcc, I8("v_cndmask_b32", ", %8, vcc"))
KERNEL(k_cmp, "v_cmp_gt_f32 vcc, %0, %8\n v_cmp_gt_f32 vcc, %1, %8\n v_cmp_gt_f32 vcc, %2, %8\n v_cmp_gt_f32 vcc, %3, %8\n"
              "v_cmp_gt_f32 vcc, %4, %8\n v_cmp_gt_f32 vcc, %5, %8\n v_cmp_gt_f32 vcc, %6, %8\n v_cmp_gt_f32 vcc, %7, %8\n")
KERNEL(k_and, I8("v_and_b32", ", %8"))
KERNEL(k_bfi, I8("v_bfi_b32", ", %8, %9"))
KERNEL(k_lshl, J8("v_lshlrev_b32", "1"))
KERNEL(k_addu, I8("v_add_u32", ", %8"))
KERNEL(k_mov, "v_mov_b32 %0, %1\n v_mov_b32 %1, %2\n v_mov_b32 %2, %3\n v_mov_b32 %3, %4\n v_mov_b32 %4, %5\n v_mov_b32 %5, %6\n v_mov_b32 %6, %7\n v_mov_b32 %7, %8\n")
KERNEL(k_cvt_i, I8("v_cvt_i32_f32", ""))
KERNEL(k_cvt_f, I8("v_cvt_f32_i32", ""))
KERNEL(k_floor, I8("v_floor_f32", ""))
KERNEL(k_fract, I8("v_fract_f32", ""))
KERNEL(k_rsq, I8("v_rsq_f32", ""))
KERNEL(k_rcp, I8("v_rcp_f32", ""))
KERNEL(k_sqrt, I8("v_sqrt_f32", ""))
KERNEL(k_exp, I8("v_exp_f32", ""))
KERNEL(k_mullo, I8("v_mul_lo_u32", ", %8"))
KERNEL(k_mad24, I8("v_mad_u32_u24", ", %8, %9"))
KERNEL(k_pkfma16, I8("v_pk_fma_f16", ", %8, %9"))
KERNEL(k_fma16, I8("v_fma_f16", ", %8, %9"))
KERNEL(k_dpp_add, I8("v_add_f32_dpp", ", %8 row_ror:1 row_mask:0xf bank_mask:0xf"))
KERNEL(k_mix_fma_max, M8("v_fma_f32", ", %8, %9", "v_max_f32", ", %8"))
KERNEL(k_mix_fma_cnd, M8("v_fma_f32", ", %8, %9", "v_cndmask_b32", ", %8, %10"))
KERNEL(k_mix_fma_rsq, M31("v_fma_f32", ", %8, %9", "v_rsq_f32", ""))
KERNEL(k_mix_mul_add, M8("v_mul_f32", ", %8", "v_add_f32", ", %9"))
KERNEL(k_mix_max_rsq, M31("v_max_f32", ", %8", "v_rsq_f32", ""))
KERNEL(k_readlane, "v_readlane_b32 s20, %0, 3\n v_readlane_b32 s21, %1, 3\n v_readlane_b32 s22, %2, 3\n v_readlane_b32 s23, %3, 3\n"
                   "v_readlane_b32 s20, %4, 3\n v_readlane_b32 s21, %5, 3\n v_readlane_b32 s22, %6, 3\n v_readlane_b32 s23, %7, 3\n")

typedef void (*kern_t)(long long*, float*, int, float, float, unsigned long long, float);

static void run(const char* name, kern_t k)
{
    const int cus = 256, iters = 20000;
    printf("%-16s", name);
    for (int occ : {1, 2, 3, 4, 8}) {
        const int wg = occ == 3 ? 768 : (occ >= 4 ? 1024 : occ * 256);      // waves per SIMD inside ONE workgroup where possible
        const int per_cu = occ == 8 ? 2 : 1;
        const int blocks = cus * per_cu;
        float* out; long long* cyc;
        (void)hipMalloc(&out, (size_t)blocks * wg * sizeof(float));
        (void)hipMalloc(&cyc, (size_t)blocks * (wg / 64) * sizeof(long long));
        hipEvent_t e0, e1;
        (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        k<<<blocks, wg>>>(cyc, out, 100, 1.0001f, 0.5f, 0x5555555555555555ull, 0.999f);
        (void)hipEventRecord(e0);
        k<<<blocks, wg>>>(cyc, out, iters, 1.0001f, 0.5f, 0x5555555555555555ull, 0.999f);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        std::vector<long long> h((size_t)blocks * (wg / 64));
        (void)hipMemcpy(h.data(), cyc, h.size() * sizeof(long long), hipMemcpyDeviceToHost);
        double sum = 0; for (long long v : h) sum += (double)v;
        const double per_wave = sum / h.size();
        const double instrs = (double)occ * iters * 8;
        // ticks of s_memtime per instruction, and nanoseconds (HIP events, includes launch overhead)
        printf(" | occ%d %6.3f tk %6.3f ns", occ, per_wave / instrs, ms * 1e6 / instrs);
        (void)hipFree(out); (void)hipFree(cyc);
    }
    printf("\n");
}

int main()
{
#define R(k) run(#k, k)
    R(k_fma); R(k_fma_mod); R(k_fma_sgpr); R(k_fmac); R(k_mul); R(k_mul_lit); R(k_mul_inl); R(k_add); R(k_sub);
    R(k_add_abs); R(k_max); R(k_min); R(k_max3); R(k_med3); R(k_cndmask_s); R(k_cndmask_vcc); R(k_cmp); R(k_and); R(k_bfi);
    R(k_lshl); R(k_addu); R(k_mov); R(k_cvt_i); R(k_cvt_f); R(k_floor); R(k_fract); R(k_rsq); R(k_rcp); R(k_sqrt); R(k_exp);
    R(k_mullo); R(k_mad24); R(k_pkfma16); R(k_fma16); R(k_dpp_add); R(k_mix_fma_max); R(k_mix_fma_cnd); R(k_mix_fma_rsq);
    R(k_mix_mul_add); R(k_mix_max_rsq); R(k_readlane);
    return 0;
}
