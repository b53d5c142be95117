// dep_latency.hip -- development microbenchmark (round 6): the latency of a DEPENDENT chain of each instruction kind on the
// critical path of one CartPole MLP rollout step, one wave per SIMD (what a small per-GPU population is: profiles/r06_small_populations.txt).
// Each kernel runs ITER x 32 instructions in which every instruction reads the previous one's result; the figure is wall time /
// instructions = issue-to-issue time of dependent instructions of a lone wave.  tools/chain_model.py prices the loop of the 16-lanes-per-env
// kernel with them (`small_shard_floor_us` on the bench line).
//   hipcc --offload-arch=gfx950 -O3 tools/dep_latency.hip -o tools/dep_latency && tools/dep_latency > profiles/r06_dep_latency.json
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

typedef float v2f __attribute__((ext_vector_type(2)));

#define R8(OP) OP OP OP OP OP OP OP OP
#define BODY32(OP) R8(OP) R8(OP) R8(OP) R8(OP)

#define KERNEL(NAME, ASM)                                                                                          \
    __global__ __launch_bounds__(256) void NAME(int iters, float seed, float *out)                                 \
    {                                                                                                              \
        __shared__ __attribute__((aligned(16))) float lds[64];                                                     \
        if (threadIdx.x < 64) lds[threadIdx.x] = 0.0f;                                                             \
        __syncthreads();                                                                                           \
        float r = seed * (1 + (threadIdx.x & 3)), a = 1.0f + seed * 1e-3f, b = seed * 1e-3f;                       \
        float q0 = 0.0f, q1 = 0.0f, q2 = 0.0f, q3 = 0.0f;                                                           \
        v2f p2 = {r, r * 0.5f}, a2 = {a, a}, b2 = {b, b};                                                            \
        (void)a2; (void)b2;                                                                                         \
        unsigned addr = (unsigned)(size_t)lds;                                                                      \
        (void)q0; (void)q1; (void)q2; (void)q3; (void)addr;                                                         \
        for (int it = 0; it < iters; ++it) { BODY32(ASM) }                                                         \
        out[blockIdx.x * blockDim.x + threadIdx.x] = r + q0 + q1 + q2 + q3 + (float)addr + p2.x + p2.y;            \
    }

#define L_FMA asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(r) : "v"(a), "v"(b));
#define L_FMAC asm volatile("v_fmac_f32_e32 %0, %1, %2" : "+v"(r) : "v"(a), "v"(b));
#define L_FMAMK asm volatile("v_fmamk_f32 %0, %0, 0x3d2aaaa5, %1" : "+v"(r) : "v"(b));
#define L_MUL asm volatile("v_mul_f32_e32 %0, %0, %1" : "+v"(r) : "v"(a));
#define L_ADD asm volatile("v_add_f32_e32 %0, %0, %1" : "+v"(r) : "v"(b));
#define L_ADD_DPP_Q asm volatile("v_add_f32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(r));
#define L_ADD_DPP_R asm volatile("v_add_f32_dpp %0, %0, %0 row_ror:8 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(r));
#define L_MOV_DPP asm volatile("v_mov_b32_dpp %0, %0 row_newbcast:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(r));
#define L_MUL_DPP asm volatile("v_mul_f32_dpp %0, %0, %1 quad_perm:[0,0,2,2] row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(r) : "v"(a));
#define L_MIN asm volatile("v_min_f32_e64 %0, |%0|, %1" : "+v"(r) : "v"(a));
#define L_CVT asm volatile("v_cvt_i32_f32_e32 %0, %0" : "+v"(r));
#define L_FRACT asm volatile("v_fract_f32_e32 %0, %0" : "+v"(r));
#define L_LSHL asm volatile("v_lshlrev_b32_e32 %0, 1, %0" : "+v"(r));
#define L_BFI asm volatile("v_bfi_b32 %0, %1, %0, %2" : "+v"(r) : "v"(a), "v"(b));
#define L_MED3 asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(r) : "v"(b), "v"(a));
#define L_RCP asm volatile("v_rcp_f32_e32 %0, %0" : "+v"(r));
// two instructions per link: the compare writes vcc, the select reads it and produces the next compare's input
#define L_CMP_CND asm volatile("v_cmp_ngt_f32_e32 vcc, %0, %1\n\tv_cndmask_b32_e32 %0, %1, %2, vcc" : "+v"(r) : "v"(a), "v"(b) : "vcc");
// compare into an SGPR pair, select from it (the e64 forms the kernel's masked loops use)
#define L_CMP_CND64 asm volatile("v_cmp_ngt_f32_e64 s[20:21], %0, %1\n\tv_cndmask_b32_e64 %0, %1, %2, s[20:21]" : "+v"(r) : "v"(a), "v"(b) : "s20", "s21");
// the table read: address from the previous result, one ds_read_b128 and the wait for it (LDS holds zeros: the address stays 0)
#define L_LDS asm volatile("v_add_u32_e32 %4, %0, %4\n\tds_read_b128 v[100:103], %4\n\ts_waitcnt lgkmcnt(0)\n\tv_mov_b32_e32 %0, v100" \
                           : "+v"(r), "+v"(q0), "+v"(q1), "+v"(q2), "+v"(addr) : : "v100", "v101", "v102", "v103");
#define L_SWAP16 asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 1\n\tv_add_f32_e32 %0, %0, %1" : "+v"(r), "+v"(q0));

#define L_PKFMA asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p2) : "v"(a2), "v"(b2));
#define L_PKMUL asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p2) : "v"(a2));
#define L_PKADD asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p2) : "v"(b2));
KERNEL(k_fma, L_FMA)
KERNEL(k_pkfma, L_PKFMA)
KERNEL(k_pkmul, L_PKMUL)
KERNEL(k_pkadd, L_PKADD)
KERNEL(k_fmac, L_FMAC)
KERNEL(k_fmamk, L_FMAMK)
KERNEL(k_mul, L_MUL)
KERNEL(k_add, L_ADD)
KERNEL(k_add_dpp_q, L_ADD_DPP_Q)
KERNEL(k_add_dpp_r, L_ADD_DPP_R)
KERNEL(k_mov_dpp, L_MOV_DPP)
KERNEL(k_mul_dpp, L_MUL_DPP)
KERNEL(k_min, L_MIN)
KERNEL(k_cvt, L_CVT)
KERNEL(k_fract, L_FRACT)
KERNEL(k_lshl, L_LSHL)
KERNEL(k_bfi, L_BFI)
KERNEL(k_med3, L_MED3)
KERNEL(k_rcp, L_RCP)
KERNEL(k_cmp_cnd, L_CMP_CND)
KERNEL(k_cmp_cnd64, L_CMP_CND64)
KERNEL(k_lds, L_LDS)
KERNEL(k_swap16, L_SWAP16)

typedef void (*kern_t)(int, float, float *);

int main()
{
    struct Case { const char *name; kern_t k; int instr_per_link; };
    const std::vector<Case> cases = {
        {"v_fma_f32", k_fma, 1}, {"v_pk_fma_f32", k_pkfma, 1}, {"v_pk_mul_f32", k_pkmul, 1}, {"v_pk_add_f32", k_pkadd, 1}, {"v_fmac_f32", k_fmac, 1}, {"v_fmamk_f32", k_fmamk, 1}, {"v_mul_f32", k_mul, 1}, {"v_add_f32", k_add, 1},
        {"v_add_f32_dpp quad_perm", k_add_dpp_q, 1}, {"v_add_f32_dpp row_ror", k_add_dpp_r, 1}, {"v_mov_b32_dpp row_newbcast", k_mov_dpp, 1},
        {"v_mul_f32_dpp quad_perm", k_mul_dpp, 1}, {"v_min_f32 |x|", k_min, 1}, {"v_cvt_i32_f32", k_cvt, 1}, {"v_fract_f32", k_fract, 1},
        {"v_lshlrev_b32", k_lshl, 1}, {"v_bfi_b32", k_bfi, 1}, {"v_med3_f32", k_med3, 1}, {"v_rcp_f32", k_rcp, 1},
        {"v_cmp vcc + v_cndmask vcc", k_cmp_cnd, 2}, {"v_cmp sgpr + v_cndmask sgpr", k_cmp_cnd64, 2},
        {"v_add_u32 + ds_read_b128 + wait + v_mov", k_lds, 3}, {"s_nop + v_permlane16_swap + s_nop + v_add", k_swap16, 2},
    };
    const int iters = 4000;
    float *out;
    CK(hipMalloc(&out, (size_t)256 * 256 * sizeof(float)));
    printf("{\n \"what\": \"ns per LINK of a dependent chain, one wave per SIMD (256 workgroups of 256 threads), wall time over links\",\n \"links\": {\n");
    bool first = true;
    for (const Case &c : cases) {
        hipLaunchKernelGGL(c.k, dim3(256), dim3(256), 0, 0, iters, 0.25f, out);
        CK(hipDeviceSynchronize());
        double best = 1e30;
        for (int rep = 0; rep < 3; ++rep) {
            hipEvent_t e0, e1;
            CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
            CK(hipEventRecord(e0, 0));
            hipLaunchKernelGGL(c.k, dim3(256), dim3(256), 0, 0, iters, 0.25f, out);
            CK(hipEventRecord(e1, 0));
            CK(hipEventSynchronize(e1));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            const double ns = (double)ms * 1e6 / ((double)iters * 32);
            if (ns < best) best = ns;
        }
        printf("%s  \"%s\": {\"ns\": %.3f, \"instructions\": %d}", first ? "" : ",\n", c.name, best, c.instr_per_link);
        first = false;
    }
    printf("\n }\n}\n");
    return 0;
}
