// valu_issue.hip -- development microbenchmark: issue cost of the VALU instruction kinds the fused rollout uses,
// per SIMD, at 1 / 2 / 4 waves per SIMD.  Each kernel runs ITER iterations of 32 instructions of one kind over 8
// independent registers (no dependency closer than 8 instructions), so the number reported is the issue cadence,
// not a latency.  Output: cycles per wave-instruction per SIMD (s_memtime ticks of the shader clock).
//   hipcc --offload-arch=gfx950 -O3 tools/valu_issue.hip -o tools/valu_issue
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

#define R8(OP) OP(0) OP(1) OP(2) OP(3) OP(4) OP(5) OP(6) OP(7)
#define BODY32(OP) R8(OP) R8(OP) R8(OP) R8(OP)

#define KERNEL(NAME, ASM)                                                                                         \
    __global__ __launch_bounds__(1024) void NAME(int iters, float seed, float *out, unsigned long long *ticks)    \
    {                                                                                                             \
        float r[8], a = seed + threadIdx.x * 1e-3f, b = 1.0f + seed;                                              \
        int ia = threadIdx.x;                                                                                     \
        for (int i = 0; i < 8; ++i) r[i] = seed * (i + 1);                                                        \
        asm volatile("s_mov_b64 s[22:23], 0x5555\n\ts_mov_b32 s24, 0x3f000000" ::: "s22", "s23", "s24");            \
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();                                               \
        for (int it = 0; it < iters; ++it) { BODY32(ASM) }                                                        \
        const unsigned long long t1 = __builtin_amdgcn_s_memtime();                                               \
        float s = 0;                                                                                              \
        for (int i = 0; i < 8; ++i) s += r[i];                                                                    \
        out[blockIdx.x * blockDim.x + threadIdx.x] = s + a + b + ia;                                              \
        if (threadIdx.x == 0) ticks[blockIdx.x] = t1 - t0;                                                        \
    }

#define A_FMA(i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(r[i]) : "v"(a), "v"(b));
#define A_FMAC(i) asm volatile("v_fmac_f32_e32 %0, %1, %2" : "+v"(r[i]) : "v"(a), "v"(b));
#define A_FMAAK(i) asm volatile("v_fmaak_f32 %0, %0, %1, 0x3d2aaaa5" : "+v"(r[i]) : "v"(a));
#define A_MUL(i) asm volatile("v_mul_f32_e32 %0, %0, %1" : "+v"(r[i]) : "v"(a));
#define A_ADD(i) asm volatile("v_add_f32_e32 %0, %0, %1" : "+v"(r[i]) : "v"(a));
#define A_MIN(i) asm volatile("v_min_f32_e64 %0, |%0|, %1" : "+v"(r[i]) : "v"(a));
#define A_MED3(i) asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(r[i]) : "v"(a), "v"(b));
#define A_CNDMASK(i) asm volatile("v_cndmask_b32_e32 %0, %0, %1, vcc" : "+v"(r[i]) : "v"(a) : );
#define A_CVT(i) asm volatile("v_cvt_i32_f32_e32 %0, %0" : "+v"(r[i]));
#define A_FRACT(i) asm volatile("v_fract_f32_e32 %0, %0" : "+v"(r[i]));
#define A_LSHL(i) asm volatile("v_lshlrev_b32_e32 %0, 4, %0" : "+v"(r[i]));
#define A_BFI(i) asm volatile("v_bfi_b32 %0, %1, %0, %2" : "+v"(r[i]) : "v"(a), "v"(b));
#define A_MOV(i) asm volatile("v_mov_b32_e32 %0, %1" : "+v"(r[i]) : "v"(a));
#define A_DPPQ(i) asm volatile("v_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(r[i]));
#define A_DPPH(i) asm volatile("v_add_f32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(r[i]));
#define A_RCP(i) asm volatile("v_rcp_f32_e32 %0, %0" : "+v"(r[i]));
#define A_DIVSCALE(i) asm volatile("v_div_scale_f32 %0, vcc, %0, %1, %0" : "+v"(r[i]) : "v"(a) : "vcc");
#define A_DIVFMAS(i) asm volatile("v_div_fmas_f32 %0, %0, %1, %2" : "+v"(r[i]) : "v"(a), "v"(b) : "vcc");
#define A_DIVFIXUP(i) asm volatile("v_div_fixup_f32 %0, %0, %1, %2" : "+v"(r[i]) : "v"(a), "v"(b));
#define A_CMP(i) asm volatile("v_cmp_gt_f32_e64 s[20:21], |%0|, %1" : : "v"(r[i]), "v"(a) : "s20", "s21");
#define A_PKFMA(i) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(d[i]) : "v"(da), "v"(db));
#define A_ADDU(i) asm volatile("v_add_u32_e32 %0, 1, %0" : "+v"(r[i]));
#define A_FMA64(i) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(d[i]) : "v"(da), "v"(db));

KERNEL(k_fma, A_FMA)
#define A_MINE32(i) asm volatile("v_min_f32_e32 %0, %0, %1" : "+v"(r[i]) : "v"(a));
#define A_MAXLIT(i) asm volatile("v_max_f32_e32 %0, 0xbf400000, %0" : "+v"(r[i]));
#define A_AND(i) asm volatile("v_and_b32_e32 %0, 0x7fffffff, %0" : "+v"(r[i]));
#define A_ANDR(i) asm volatile("v_and_b32_e32 %0, %1, %0" : "+v"(r[i]) : "v"(a));
#define A_SUB(i) asm volatile("v_sub_f32_e32 %0, %0, %1" : "+v"(r[i]) : "v"(a));
#define A_LSHLADD(i) asm volatile("v_lshl_add_u32 %0, %0, 4, %1" : "+v"(r[i]) : "v"(a));
#define A_MADU24(i) asm volatile("v_mad_u32_u24 %0, %0, 16, %1" : "+v"(r[i]) : "v"(a));
#define A_MULU24(i) asm volatile("v_mul_u32_u24_e32 %0, 16, %0" : "+v"(r[i]));
#define A_CVTU(i) asm volatile("v_cvt_u32_f32_e32 %0, %0" : "+v"(r[i]));
#define A_FLOOR(i) asm volatile("v_floor_f32_e32 %0, %0" : "+v"(r[i]));
#define A_RNDNE(i) asm volatile("v_rndne_f32_e32 %0, %0" : "+v"(r[i]));
#define A_LSHR(i) asm volatile("v_lshrrev_b32_e32 %0, 4, %0" : "+v"(r[i]));
#define A_CNDE64(i) asm volatile("v_cndmask_b32_e64 %0, %0, %1, s[22:23]" : "+v"(r[i]) : "v"(a));
#define A_FMA2SRC(i) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(r[i]) : "v"(a));
#define A_FMACONST(i) asm volatile("v_fma_f32 %0, %0, %1, 1.0" : "+v"(r[i]) : "v"(a));
#define A_FMASGPR(i) asm volatile("v_fma_f32 %0, %0, %1, s24" : "+v"(r[i]) : "v"(a));
#define A_MULE64(i) asm volatile("v_mul_f32_e64 %0, %0, %1" : "+v"(r[i]) : "v"(a));
#define A_MULABS(i) asm volatile("v_mul_f32_e64 %0, |%0|, %1" : "+v"(r[i]) : "v"(a));
#define A_FMAMK(i) asm volatile("v_fmamk_f32 %0, %0, 0x3d2aaaa5, %1" : "+v"(r[i]) : "v"(a));
#define A_MULLIT(i) asm volatile("v_mul_f32_e32 %0, 0x42000000, %0" : "+v"(r[i]));
#define A_ADDCO(i) asm volatile("v_addc_co_u32_e64 %0, vcc, %0, 0, s[22:23]" : "+v"(r[i]) : : "vcc");
KERNEL(k_mine32, A_MINE32)
KERNEL(k_maxlit, A_MAXLIT)
KERNEL(k_and, A_AND)
KERNEL(k_andr, A_ANDR)
KERNEL(k_sub, A_SUB)
KERNEL(k_lshladd, A_LSHLADD)
KERNEL(k_madu24, A_MADU24)
KERNEL(k_mulu24, A_MULU24)
KERNEL(k_cvtu, A_CVTU)
KERNEL(k_floor, A_FLOOR)
KERNEL(k_rndne, A_RNDNE)
KERNEL(k_lshr, A_LSHR)
KERNEL(k_cnde64, A_CNDE64)
KERNEL(k_fma2src, A_FMA2SRC)
KERNEL(k_fmaconst, A_FMACONST)
KERNEL(k_fmasgpr, A_FMASGPR)
KERNEL(k_mule64, A_MULE64)
KERNEL(k_mulabs, A_MULABS)
KERNEL(k_fmamk, A_FMAMK)
KERNEL(k_mullit, A_MULLIT)
KERNEL(k_addco, A_ADDCO)
KERNEL(k_fmac, A_FMAC)
KERNEL(k_fmaak, A_FMAAK)
KERNEL(k_mul, A_MUL)
KERNEL(k_add, A_ADD)
KERNEL(k_min, A_MIN)
KERNEL(k_med3, A_MED3)
KERNEL(k_cndmask, A_CNDMASK)
KERNEL(k_cvt, A_CVT)
KERNEL(k_fract, A_FRACT)
KERNEL(k_lshl, A_LSHL)
KERNEL(k_bfi, A_BFI)
KERNEL(k_mov, A_MOV)
KERNEL(k_dppq, A_DPPQ)
KERNEL(k_dpph, A_DPPH)
KERNEL(k_rcp, A_RCP)
KERNEL(k_divscale, A_DIVSCALE)
KERNEL(k_divfmas, A_DIVFMAS)
KERNEL(k_divfixup, A_DIVFIXUP)
KERNEL(k_cmp, A_CMP)
KERNEL(k_addu, A_ADDU)

// packed / double kinds need 64-bit registers
#define KERNEL64(NAME, ASM)                                                                                       \
    __global__ __launch_bounds__(1024) void NAME(int iters, float seed, float *out, unsigned long long *ticks)    \
    {                                                                                                             \
        double d[8], da = seed + threadIdx.x * 1e-3, db = 1.0 + seed;                                             \
        for (int i = 0; i < 8; ++i) d[i] = seed * (i + 1);                                                        \
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();                                               \
        for (int it = 0; it < iters; ++it) { BODY32(ASM) }                                                        \
        const unsigned long long t1 = __builtin_amdgcn_s_memtime();                                               \
        double s = 0;                                                                                             \
        for (int i = 0; i < 8; ++i) s += d[i];                                                                    \
        out[blockIdx.x * blockDim.x + threadIdx.x] = (float)(s + da + db);                                        \
        if (threadIdx.x == 0) ticks[blockIdx.x] = t1 - t0;                                                        \
    }
KERNEL64(k_pkfma, A_PKFMA)
#define A_PKFMABC(i) asm volatile("v_pk_fma_f32 %0, %0, %1, %2 op_sel_hi:[1,0,1]" : "+v"(d[i]) : "v"(da), "v"(db));
KERNEL64(k_pkfma_bc, A_PKFMABC)
#define A_PKMUL(i) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(d[i]) : "v"(da));
KERNEL64(k_pkmul, A_PKMUL)
#define A_PKADD(i) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(d[i]) : "v"(da));
KERNEL64(k_pkadd, A_PKADD)
#define A_PKFMAACC(i) asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(d[i]) : "v"(da), "v"(db));
KERNEL64(k_pkfma_acc, A_PKFMAACC)
KERNEL64(k_fma64, A_FMA64)

// a dependent chain: every instruction reads the previous result (latency, one wave per SIMD tells the story)
#define A_FMADEP(i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(r[0]) : "v"(a), "v"(b));
KERNEL(k_fma_dep, A_FMADEP)
#define A_FMADEP2(i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(r[i & 1]) : "v"(a), "v"(b));
KERNEL(k_fma_dep2, A_FMADEP2)

typedef void (*kern_t)(int, float, float *, unsigned long long *);

int main()
{
    struct Case { const char *name; kern_t k; };
    const std::vector<Case> cases = {
        {"v_fma_f32 (vop3)", k_fma}, {"v_fmac_f32 e32", k_fmac}, {"v_fmaak_f32 literal", k_fmaak}, {"v_mul_f32", k_mul},
        {"v_add_f32", k_add}, {"v_min_f32 |x| e64", k_min}, {"v_med3_f32", k_med3}, {"v_cndmask_b32 vcc", k_cndmask},
        {"v_cvt_i32_f32", k_cvt}, {"v_fract_f32", k_fract}, {"v_lshlrev_b32", k_lshl}, {"v_bfi_b32", k_bfi},
        {"v_mov_b32", k_mov}, {"v_add_u32", k_addu}, {"v_add_f32 dpp quad_perm", k_dppq}, {"v_add_f32 dpp row_half_mirror", k_dpph},
        {"v_rcp_f32", k_rcp}, {"v_div_scale_f32", k_divscale}, {"v_div_fmas_f32", k_divfmas}, {"v_div_fixup_f32", k_divfixup},
        {"v_cmp_gt_f32 -> sgpr", k_cmp}, {"v_pk_fma_f32", k_pkfma}, {"v_fma_f64", k_fma64},
        {"v_min_f32 e32", k_mine32}, {"v_max_f32 literal", k_maxlit}, {"v_and_b32 literal", k_and}, {"v_and_b32 reg", k_andr},
        {"v_sub_f32", k_sub}, {"v_lshl_add_u32", k_lshladd}, {"v_mad_u32_u24", k_madu24}, {"v_mul_u32_u24", k_mulu24},
        {"v_cvt_u32_f32", k_cvtu}, {"v_floor_f32", k_floor}, {"v_rndne_f32", k_rndne}, {"v_lshrrev_b32", k_lshr},
        {"v_cndmask_b32 e64 sgpr mask", k_cnde64}, {"v_fma_f32 r,r,a,a (2 distinct)", k_fma2src},
        {"v_fma_f32 r,r,a,1.0", k_fmaconst}, {"v_fma_f32 r,r,a,s24", k_fmasgpr}, {"v_mul_f32 e64", k_mule64},
        {"v_mul_f32 e64 |x|", k_mulabs}, {"v_fmamk_f32 literal", k_fmamk}, {"v_mul_f32 literal", k_mullit},
        {"v_addc_co_u32 e64", k_addco},
        {"v_pk_fma_f32 op_sel broadcast", k_pkfma_bc}, {"v_pk_fma_f32 acc form broadcast", k_pkfma_acc},
        {"v_pk_mul_f32", k_pkmul}, {"v_pk_add_f32", k_pkadd},
        {"v_fma_f32 dependent chain", k_fma_dep}, {"v_fma_f32 two interleaved chains", k_fma_dep2},
    };
    const int iters = 2000;
    float *out;
    unsigned long long *ticks;
    CK(hipMalloc(&out, (size_t)512 * 1024 * 4 * sizeof(float)));
    CK(hipMalloc(&ticks, 512 * 4 * sizeof(unsigned long long)));
    std::vector<unsigned long long> h(512 * 4);
    printf("%-34s %10s %10s %10s %10s   (ns per wave-instruction per SIMD, by waves per SIMD)\n", "instruction", "1 wave", "2 waves", "4 waves", "8 waves");
    for (const Case &c : cases) {
        printf("%-34s", c.name);
        for (int wps : {1, 2, 4, 8}) {
            const int block = wps == 8 ? 1024 : 256 * wps;     // wps waves on each of the CU's 4 SIMDs
            const int n_cu = wps == 8 ? 512 : 256;             // 8: two 1024-thread workgroups per CU
            hipLaunchKernelGGL(c.k, dim3(n_cu), dim3(block), 0, 0, iters, 0.25f, out, ticks);
            CK(hipDeviceSynchronize());
            hipLaunchKernelGGL(c.k, dim3(n_cu), dim3(block), 0, 0, iters, 0.25f, out, ticks);
            CK(hipDeviceSynchronize());
            CK(hipMemcpy(h.data(), ticks, n_cu * sizeof(unsigned long long), hipMemcpyDeviceToHost));
            double sum = 0;
            for (int i = 0; i < n_cu; ++i) sum += (double)h[i];
            // s_memtime counts at a fixed 100 MHz on gfx9; convert with the measured wall clock instead
            hipEvent_t e0, e1;
            CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
            CK(hipEventRecord(e0, 0));
            hipLaunchKernelGGL(c.k, dim3(n_cu), dim3(block), 0, 0, iters * 4, 0.25f, out, ticks);
            CK(hipEventRecord(e1, 0));
            CK(hipEventSynchronize(e1));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            const double ns_per_instr = (double)ms * 1e6 / ((double)iters * 4 * 32 * wps);   // 512 x 1024 = 8 per SIMD too
            printf(" %7.2f ns", ns_per_instr);
            (void)sum;
        }
        printf("\n");
    }
    printf("(ns per wave-instruction per SIMD; multiply by the clock in GHz for cycles: 2.4 GHz -> 2 cycles = 0.83 ns)\n");
    return 0;
}
