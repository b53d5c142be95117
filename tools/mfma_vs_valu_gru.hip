// mfma_vs_valu_gru.hip -- development microbenchmark behind the "no MFMA for the policy contraction" decision
// (DESIGN.md section 4).  The GRU gate contraction of ONE offspring is G[96 x E] = W[96 x 32] . H[32 x E]
// (E = episodes evaluated together, 5 in the reference's default).  Two wave-per-offspring kernels with the weights
// resident in VGPRs are timed over T recurrent steps:
//   valu : lane = (hidden unit j, k-half); 3 gates x 16 fma per episode, v_permlane32_swap half-sum  (production form)
//   mfma : v_mfma_f32_16x16x4_f32, A = W fragments in VGPRs, B = H padded to 16 columns, 6 row tiles x 8 k-steps
// Both compute h'[u][e] = 0.55 * (G_r + G_z + G_n)[u][e]; results are compared after 6 steps (the two forms sum in
// different orders: VALU (k<16)+(k>=16), MFMA k-ascending), timing uses 200 steps.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fno-slp-vectorize tools/mfma_vs_valu_gru.hip -o tools/mfma_vs_valu_gru
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float half_pair_sum(float x)
{
    float a = x, b = x;
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
    return a + b;
}

// W: [n_off][96][32]; h0: [n_off][32][EP]; out: [n_off][32][EP]
template <int EP>
__global__ __launch_bounds__(256) void k_valu(const float *__restrict__ W, const float *__restrict__ h0, int n_off, int T,
                                              float *__restrict__ out)
{
    __shared__ __attribute__((aligned(16))) float hv[4][EP][32];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    int o = blockIdx.x * 4 + wave;
    const bool valid = o < n_off;
    o = valid ? o : n_off - 1;
    const int j = lane & 31, kh = lane >> 5;
    float w[3][16];
#pragma unroll
    for (int g = 0; g < 3; ++g)
#pragma unroll
        for (int k = 0; k < 16; ++k) w[g][k] = W[((size_t)o * 96 + g * 32 + j) * 32 + 16 * kh + k];
    if (kh == 0)
        for (int e = 0; e < EP; ++e) hv[wave][e][j] = h0[((size_t)o * 32 + j) * EP + e];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    for (int t = 0; t < T; ++t) {
        float hn[EP];
#pragma unroll
        for (int e = 0; e < EP; ++e) {
            float acc[3] = {0.0f, 0.0f, 0.0f};
            const float4 *v = reinterpret_cast<const float4 *>(&hv[wave][e][16 * kh]);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 x = v[q];
                const float xe[4] = {x.x, x.y, x.z, x.w};
#pragma unroll
                for (int c = 0; c < 4; ++c)
#pragma unroll
                    for (int g = 0; g < 3; ++g) acc[g] = __builtin_fmaf(w[g][4 * q + c], xe[c], acc[g]);
            }
            const float s = (half_pair_sum(acc[0]) + half_pair_sum(acc[1])) + half_pair_sum(acc[2]);
            hn[e] = 0.55f * s;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        if (kh == 0)
#pragma unroll
            for (int e = 0; e < EP; ++e) hv[wave][e][j] = hn[e];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
    if (valid && kh == 0)
        for (int e = 0; e < EP; ++e) out[((size_t)o * 32 + j) * EP + e] = hv[wave][e][j];
}

// MFMA form: D[16x16] += A[16x4] . B[4x16]; A lane l: A[i = l&15][k = l>>4]; B lane l: B[k = l>>4][j = l&15];
// D reg r of lane l: D[row = 4*(l>>4) + r][col = l&15].
template <int EP>
__global__ __launch_bounds__(256) void k_mfma(const float *__restrict__ W, const float *__restrict__ h0, int n_off, int T,
                                              float *__restrict__ out)
{
    __shared__ float hv[4][32][16];                       // [unit][column], columns >= EP are zero padding
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    int o = blockIdx.x * 4 + wave;
    const bool valid = o < n_off;
    o = valid ? o : n_off - 1;
    const int li = lane & 15, lg = lane >> 4;
    float a[6][8];                                         // 6 row tiles x 8 k-steps
#pragma unroll
    for (int m = 0; m < 6; ++m)
#pragma unroll
        for (int s = 0; s < 8; ++s) a[m][s] = W[((size_t)o * 96 + 16 * m + li) * 32 + 4 * s + lg];
    for (int idx = lane; idx < 32 * 16; idx += 64) {
        const int u = idx >> 4, c = idx & 15;
        hv[wave][u][c] = c < EP ? h0[((size_t)o * 32 + u) * EP + c] : 0.0f;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    for (int t = 0; t < T; ++t) {
        float b[8];
#pragma unroll
        for (int s = 0; s < 8; ++s) b[s] = hv[wave][4 * s + lg][li];
        f32x4 d[6];
#pragma unroll
        for (int m = 0; m < 6; ++m) {
            d[m] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
            for (int s = 0; s < 8; ++s) d[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[m][s], b[s], d[m], 0, 0, 0);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        // unit u = 16*half + 4*lg + r lives in tiles (half, half + 2, half + 4) = gates r, z, n
#pragma unroll
        for (int half = 0; half < 2; ++half)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                hv[wave][16 * half + 4 * lg + r][li] = 0.55f * ((d[half][r] + d[half + 2][r]) + d[half + 4][r]);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
    if (valid)
        for (int idx = lane; idx < 32 * 16; idx += 64) {
            const int u = idx >> 4, c = idx & 15;
            if (c < EP) out[((size_t)o * 32 + u) * EP + c] = hv[wave][u][c];
        }
}

template <typename F>
static float time_ms(F launch, int reps = 5)
{
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    launch();
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0, 0));
    for (int i = 0; i < reps; ++i) launch();
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    return ms / reps;
}

template <int EP>
static void run(int n_off, int T)
{
    std::vector<float> W((size_t)n_off * 96 * 32), h((size_t)n_off * 32 * EP);
    unsigned s = 12345;
    auto rnd = [&] { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xffff) / 65536.0f - 0.5f; };
    for (auto &v : W) v = rnd() * 0.6f;
    for (auto &v : h) v = rnd();
    float *dW, *dh, *o1, *o2;
    CK(hipMalloc(&dW, W.size() * 4)); CK(hipMalloc(&dh, h.size() * 4)); CK(hipMalloc(&o1, h.size() * 4)); CK(hipMalloc(&o2, h.size() * 4));
    CK(hipMemcpy(dW, W.data(), W.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dh, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    const int blocks = (n_off + 3) / 4;
    const float tv = time_ms([&] { hipLaunchKernelGGL((k_valu<EP>), dim3(blocks), dim3(256), 0, 0, dW, dh, n_off, T, o1); });
    const float tm = time_ms([&] { hipLaunchKernelGGL((k_mfma<EP>), dim3(blocks), dim3(256), 0, 0, dW, dh, n_off, T, o2); });
    hipLaunchKernelGGL((k_valu<EP>), dim3(blocks), dim3(256), 0, 0, dW, dh, n_off, 6, o1);
    hipLaunchKernelGGL((k_mfma<EP>), dim3(blocks), dim3(256), 0, 0, dW, dh, n_off, 6, o2);
    CK(hipDeviceSynchronize());
    std::vector<float> r1(h.size()), r2(h.size());
    CK(hipMemcpy(r1.data(), o1, h.size() * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(r2.data(), o2, h.size() * 4, hipMemcpyDeviceToHost));
    double md = 0, mx = 0;
    for (size_t i = 0; i < r1.size(); ++i) { md = fmax(md, fabs((double)r1[i] - r2[i])); mx = fmax(mx, fabs((double)r1[i])); }
    const double steps = (double)n_off * EP * T;
    printf("E = %2d, %d offspring, %d steps: VALU %8.3f ms (%6.2f ns/env-step)   MFMA 16x16x4 %8.3f ms (%6.2f ns/env-step)   "
           "MFMA/VALU time = %.2f   max|diff| = %.2e (max|h| %.2e)\n",
           EP, n_off, T, tv, tv * 1e6 / steps, tm, tm * 1e6 / steps, tm / tv, md, mx);
    hipFree(dW); hipFree(dh); hipFree(o1); hipFree(o2);
}

int main()
{
    for (int n_off : {4096, 16384}) {
        run<5>(n_off, 200);
        run<8>(n_off, 200);
        run<16>(n_off, 200);
    }
    return 0;
}
