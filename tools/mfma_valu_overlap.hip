// mfma_valu_overlap.hip -- does a SIMD of gfx950 run one wave's v_mfma_f32_4x4x1_16b_f32 stream UNDER another wave's VALU stream?
// (round 6, after k_rollout_gru_mfma4 measured ~3560 cycles per wave-step with two waves per SIMD: 196 MFMA x 8 + 366 VALU x 4 + LDS,
// i.e. the SUM of its phases, as if nothing overlapped.)
//
// One workgroup per CU (the dynamic LDS request keeps a second one away).  Streams per wave and trip:
//   M : 12 independent accumulation chains of v_mfma_f32_4x4x1_16b_f32 (the kernel's gate chains)
//   V : 24 independent v_fma_f32 chains
// Cases:
//   m4 / v4      4 waves per CU, one per SIMD: the lone-wave rate of each stream
//   mm / vv      8 waves per CU, two per SIMD, both the same stream
//   mv           8 waves per CU: waves 0-3 run M, waves 4-7 run V (the SIMD of each wave is read from HW_ID and printed)
//   x4 / mx      the same with an INTEGER stream X (12 chains of v_lshrrev_b32 + v_xor_b32) beside M: is it the fp32 datapath or the issue?
//   i1 / i2      ONE wave per SIMD running M and V interleaved in its own stream (1 MFMA : 2 VALU, by sched_group_barrier); i2: two such waves
// If the matrix pipe runs under the VALU, mv costs max(m, v); if they share the issue, m + v.
//   hipcc --offload-arch=gfx950 -O3 -fno-slp-vectorize tools/mfma_valu_overlap.hip -o tools/mfma_valu_overlap && tools/mfma_valu_overlap
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void stream_m(int trips, float a, float b, f32x4 (&acc)[12])
{
    for (int t = 0; t < trips; ++t) {
#pragma unroll
        for (int c = 0; c < 12; ++c) acc[c] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, acc[c], 0, 0, 0);
    }
}

__device__ __forceinline__ void stream_v(int trips, float a, float b, float (&acc)[24])
{
    for (int t = 0; t < trips; ++t) {
#pragma unroll
        for (int c = 0; c < 24; ++c) acc[c] = __builtin_fmaf(acc[c], a, b);
    }
}

__device__ __forceinline__ void stream_x(int trips, unsigned (&acc)[12])
{
    for (int t = 0; t < trips; ++t) {
#pragma unroll
        for (int c = 0; c < 12; ++c) acc[c] = acc[c] ^ (acc[c] >> 1);
    }
}

__device__ __forceinline__ void stream_i(int trips, float a, float b, f32x4 (&m)[12], float (&v)[24])
{
    for (int t = 0; t < trips; ++t) {
#pragma unroll
        for (int c = 0; c < 12; ++c) {
            m[c] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, m[c], 0, 0, 0);
            v[2 * c] = __builtin_fmaf(v[2 * c], a, b);
            v[2 * c + 1] = __builtin_fmaf(v[2 * c + 1], a, b);
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);      // one MFMA
            __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);      // two VALU
        }
    }
}

// mode 0: every wave M; 1: every wave V; 2: waves 0-3 M, the others V; 3: every wave interleaved; 4: every wave X; 5: waves 0-3 M, the others X
__global__ __launch_bounds__(512) void k_overlap(int mode, int trips, const float *__restrict__ in, float *__restrict__ out, int *__restrict__ simd_of)
{
    const int wave = threadIdx.x >> 6;
    const float a = in[threadIdx.x & 63], b = in[64 + (threadIdx.x & 63)];
    f32x4 m[12];
    float v[24];
#pragma unroll
    for (int c = 0; c < 12; ++c) m[c] = f32x4{a, b, a, b} * (float)(c + 1);
#pragma unroll
    for (int c = 0; c < 24; ++c) v[c] = a * (float)(c + 1);
    unsigned x[12];
#pragma unroll
    for (int c = 0; c < 12; ++c) x[c] = __float_as_uint(a) * (unsigned)(2 * c + 1);
    const bool do_m = mode == 0 || ((mode == 2 || mode == 5) && wave < 4);
    if (mode == 3) stream_i(trips, a, b, m, v);
    else if (mode >= 4 && !do_m) stream_x(trips, x);
    else if (do_m) stream_m(trips, a, b, m);
    else stream_v(trips, a, b, v);
    float s = 0.0f;
#pragma unroll
    for (int c = 0; c < 12; ++c) s += m[c][0] + m[c][1] + m[c][2] + m[c][3];
#pragma unroll
    for (int c = 0; c < 24; ++c) s += v[c];
#pragma unroll
    for (int c = 0; c < 12; ++c) s += (float)(x[c] & 255u);
    out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (blockIdx.x == 0 && (threadIdx.x & 63) == 0)
        simd_of[wave] = (int)__builtin_amdgcn_s_getreg((1 << 11) | (4 << 6) | 4);     // HW_REG_HW_ID, bits [5:4]: SIMD_ID
}

static float run(int mode, int waves, int trips, const float *in, float *out, int *simd_of)
{
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const size_t lds = 100 * 1024;                                   // one workgroup per CU
    for (int rep = 0; rep < 2; ++rep) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(k_overlap, dim3(256), dim3(64 * waves), lds, 0, mode, trips, in, out, simd_of);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
    }
    float ms = 0.0f;
    CK(hipEventElapsedTime(&ms, e0, e1));
    return ms;
}

int main()
{
    float *in, *out;
    int *simd_of;
    CK(hipMalloc(&in, 128 * 4)); CK(hipMalloc(&out, 256 * 512 * 4)); CK(hipMalloc(&simd_of, 8 * 4));
    std::vector<float> h(128);
    for (int i = 0; i < 128; ++i) h[i] = 0.5f + 0.001f * i;
    CK(hipMemcpy(in, h.data(), 128 * 4, hipMemcpyHostToDevice));
    CK(hipFuncSetAttribute((const void *)k_overlap, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024));
    int clk_khz = 0;
    CK(hipDeviceGetAttribute(&clk_khz, hipDeviceAttributeClockRate, 0));
    const int trips = 20000;
    const double n_m = 12.0 * trips, n_v = 24.0 * trips;
    struct { const char *name; int mode, waves; } cases[] = {{"m4", 0, 4}, {"v4", 1, 4}, {"mm", 0, 8}, {"vv", 1, 8}, {"mv", 2, 8}, {"x4", 4, 4}, {"mx", 5, 8}, {"i1", 3, 4}, {"i2", 3, 8}};
    double t[9];
    for (int c = 0; c < 9; ++c) {
        t[c] = run(cases[c].mode, cases[c].waves, trips, in, out, simd_of) * 1e-3;
        int simd[8];
        CK(hipMemcpy(simd, simd_of, sizeof simd, hipMemcpyDeviceToHost));
        printf("%s: %8.1f us", cases[c].name, t[c] * 1e6);
        if (cases[c].mode == 0) printf("   %.2f cycles per MFMA and SIMD", t[c] * clk_khz * 1e3 / (n_m * cases[c].waves / 4));
        if (cases[c].mode == 1) printf("   %.2f cycles per VALU and SIMD", t[c] * clk_khz * 1e3 / (n_v * cases[c].waves / 4));
        if (cases[c].mode == 2) printf("   m4 + v4 = %.1f us, max = %.1f us", (t[0] + t[1]) * 1e6, (t[0] > t[1] ? t[0] : t[1]) * 1e6);
        if (cases[c].mode == 4) printf("   %.2f cycles per VALU and SIMD", t[c] * clk_khz * 1e3 / (n_v * cases[c].waves / 4));
        if (cases[c].mode == 5) printf("   m4 + x4 = %.1f us, max = %.1f us", (t[0] + t[5]) * 1e6, (t[0] > t[5] ? t[0] : t[5]) * 1e6);
        if (cases[c].mode == 3) printf("   %.2f cycles per (1 MFMA + 2 VALU) and SIMD", t[c] * clk_khz * 1e3 / (n_m * cases[c].waves / 4));
        printf("   SIMD of waves:");
        for (int w = 0; w < cases[c].waves; ++w) printf(" %d", (simd[w] >> 0) & 3);
        printf("\n");
    }
    printf("clock %d MHz\n", clk_khz / 1000);
    return 0;
}
