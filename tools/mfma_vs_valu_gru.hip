// mfma_vs_valu_gru.hip -- development microbenchmark behind the "no MFMA for the policy contraction" decision
// (DESIGN.md section 4).  The GRU gate contraction of ONE offspring is G[96 x E] = W[96 x 32] . H[32 x E]
// (E = episodes evaluated together, 5 in the reference's default).  Two wave-per-offspring kernels with the weights
// resident in VGPRs are timed over T recurrent steps:
//   valu : lane = (hidden unit j, k-half); 3 gates x 16 fma per episode, v_permlane32_swap half-sum  (production form)
//   mfma : v_mfma_f32_16x16x4_f32, A = W fragments in VGPRs, B = H padded to 16 columns, 6 row tiles x 8 k-steps
//   mfma4: v_mfma_f32_4x4x1_16b_f32 (round 6): 16 INDEPENDENT 4x4x1 blocks per instruction, each with its own A (4 gate rows) and
//          B (4 episodes): columns are padded to a multiple of 4, not of 16 -- at E = 5 two column blocks (5 of 8 columns used)
//          x 24 row blocks = 48 blocks = 3 instructions per k, 96 per step; one k per instruction, so the accumulation is the
//          k-ascending fma chain by construction (checked bitwise against fmaf on the host)
// Both compute h'[u][e] = 0.55 * (G_r + G_z + G_n)[u][e]; results are compared after 6 steps (the two forms sum in
// different orders: VALU (k<16)+(k>=16), MFMA k-ascending), timing uses 200 steps.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fno-slp-vectorize tools/mfma_vs_valu_gru.hip -o tools/mfma_vs_valu_gru
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
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


// v_mfma_f32_4x4x1_16b_f32: D_b[4x4] += A_b[4x1] . B_b[1x4] for 16 blocks b.  Lane l: block b = l >> 2, i = l & 3;
// A operand = A_b[i], B operand = B_b[i], D register r = D_b[r][i].
// Block slot b of instruction m covers gate g, unit rows 4 * ub .. 4 * ub + 3, episodes 4 * cb .. 4 * cb + 3 with
//   CB = 2 (E 5..8):  cb = b & 1, ub = b >> 1, g = m                       (3 instructions per k: one gate each)
//   CB = 4 (E 9..16): cb = b & 3, ub = 4 * (m & 1) + (b >> 2), g = m >> 1  (6 instructions per k)
//   CB = 1 (E 1..4):  cb = 0, ub = b & 7, g = 2 * m + (b >> 3); g = 3 does not exist: 2 instructions per k, the second half empty
// so that the three gates of a (unit, episode) end up in one lane (CB = 1: gate z crosses from the upper half of the wave
// with one v_permlane32_swap per accumulator register).
template <int EP>
__global__ __launch_bounds__(256) void k_mfma4(const float *__restrict__ W, const float *__restrict__ h0, int n_off, int T,
                                               float *__restrict__ out)
{
    constexpr int CB = EP <= 4 ? 1 : (EP <= 8 ? 2 : 4);
    constexpr int NI = CB == 1 ? 2 : (CB == 2 ? 3 : 6);
    __shared__ __attribute__((aligned(16))) float hv[4][32][4 * CB];   // [unit = k][episode], padded columns are zero
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    int o = blockIdx.x * 4 + wave;
    const bool valid = o < n_off;
    o = valid ? o : n_off - 1;
    const int b = lane >> 2, i = lane & 3;
    const int cb = CB == 1 ? 0 : (CB == 2 ? (b & 1) : (b & 3));
    float a[NI][32];
#pragma unroll
    for (int m = 0; m < NI; ++m) {
        const int g = CB == 1 ? 2 * m + (b >> 3) : (CB == 2 ? m : (m >> 1));
        const int ub = CB == 1 ? (b & 7) : (CB == 2 ? (b >> 1) : 4 * (m & 1) + (b >> 2));
#pragma unroll
        for (int k = 0; k < 32; ++k) a[m][k] = g < 3 ? W[((size_t)o * 96 + 32 * g + 4 * ub + i) * 32 + k] : 0.0f;
    }
    for (int idx = lane; idx < 32 * 4 * CB; idx += 64) {
        const int u = idx / (4 * CB), c = idx % (4 * CB);
        hv[wave][u][c] = c < EP ? h0[((size_t)o * 32 + u) * EP + c] : 0.0f;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    for (int t = 0; t < T; ++t) {
        float bv[32];
#pragma unroll
        for (int k = 0; k < 32; ++k) bv[k] = hv[wave][k][4 * cb + i];
        f32x4 d[NI];
#pragma unroll
        for (int m = 0; m < NI; ++m) d[m] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
        for (int k = 0; k < 32; ++k)
#pragma unroll
            for (int m = 0; m < NI; ++m) d[m] = __builtin_amdgcn_mfma_f32_4x4x1f32(a[m][k], bv[k], d[m], 0, 0, 0);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        if constexpr (CB == 2) {
            const int ub = b >> 1;
#pragma unroll
            for (int r = 0; r < 4; ++r) hv[wave][4 * ub + r][4 * cb + i] = 0.55f * ((d[0][r] + d[1][r]) + d[2][r]);
        } else if constexpr (CB == 4) {
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                const int ub = 4 * half + (b >> 2);
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    hv[wave][4 * ub + r][4 * cb + i] = 0.55f * ((d[half][r] + d[half + 2][r]) + d[half + 4][r]);
            }
        } else {
            // lower half of the wave: d[0] = gate r, d[1] = gate n; upper half: d[0] = gate z (d[1] empty)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float lo = d[0][r], hi = d[0][r];
                asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(lo), "+v"(hi));   // lo: (r | r), hi: (z | z)
                if (b < 8) hv[wave][4 * (b & 7) + r][i] = 0.55f * ((lo + hi) + d[1][r]);
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
    if (valid)
        for (int idx = lane; idx < 32 * 4 * CB; idx += 64) {
            const int u = idx / (4 * CB), c = idx % (4 * CB);
            if (c < EP) out[((size_t)o * 32 + u) * EP + c] = hv[wave][u][c];
        }
}

// The PRODUCTION form of the VALU contraction (csrc/ses_gru_lockstep.h): the input side (W_ih . a) and the hidden side (W_hh . h) of a
// gate row ride ONE v_pk_fma_f32 -- weights as (W_ih, W_hh) register pairs, operands as (a_k, h_k) pairs from LDS -- so a step costs the
// 48 packed instructions per episode that k_valu spends on the hidden side alone.  Here: a = 0.5 h, W_ih[row][k] = W[row][(k + 1) & 31],
// h' = 0.55 ((Gr + Gz) + Gn) + 0.05 ((Ir + Iz) + In).
typedef float v2f __attribute__((ext_vector_type(2)));
template <int EP>
__global__ __launch_bounds__(256) void k_valu_pk(const float *__restrict__ W, const float *__restrict__ h0, int n_off, int T,
                                                 float *__restrict__ out)
{
    __shared__ __attribute__((aligned(16))) float hv[4][EP][32][2];          // (a_k, h_k) pairs
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    int o = blockIdx.x * 4 + wave;
    const bool valid = o < n_off;
    o = valid ? o : n_off - 1;
    const int j = lane & 31, kh = lane >> 5;
    v2f w[3][16];
#pragma unroll
    for (int g = 0; g < 3; ++g)
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const int kk = 16 * kh + k;
            w[g][k] = v2f{W[((size_t)o * 96 + g * 32 + j) * 32 + ((kk + 1) & 31)], W[((size_t)o * 96 + g * 32 + j) * 32 + kk]};
        }
    if (kh == 0)
        for (int e = 0; e < EP; ++e) {
            const float h = h0[((size_t)o * 32 + j) * EP + e];
            hv[wave][e][j][0] = 0.5f * h;
            hv[wave][e][j][1] = h;
        }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    for (int t = 0; t < T; ++t) {
        float hn[EP];
#pragma unroll
        for (int e = 0; e < EP; ++e) {
            v2f acc[3] = {v2f{0.0f, 0.0f}, v2f{0.0f, 0.0f}, v2f{0.0f, 0.0f}};
            const float4 *v = reinterpret_cast<const float4 *>(&hv[wave][e][16 * kh][0]);
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const float4 x = v[q];
                const v2f x0 = {x.x, x.y}, x1 = {x.z, x.w};
#pragma unroll
                for (int g = 0; g < 3; ++g) acc[g] = __builtin_elementwise_fma(w[g][2 * q], x0, acc[g]);
#pragma unroll
                for (int g = 0; g < 3; ++g) acc[g] = __builtin_elementwise_fma(w[g][2 * q + 1], x1, acc[g]);
            }
            const float si = (half_pair_sum(acc[0].x) + half_pair_sum(acc[1].x)) + half_pair_sum(acc[2].x);
            const float sh = (half_pair_sum(acc[0].y) + half_pair_sum(acc[1].y)) + half_pair_sum(acc[2].y);
            hn[e] = 0.55f * sh + 0.05f * si;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        if (kh == 0)
#pragma unroll
            for (int e = 0; e < EP; ++e) {
                hv[wave][e][j][0] = 0.5f * hn[e];
                hv[wave][e][j][1] = hn[e];
            }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
    if (valid && kh == 0)
        for (int e = 0; e < EP; ++e) out[((size_t)o * 32 + j) * EP + e] = hv[wave][e][j][1];
}

// Both sides on v_mfma_f32_4x4x1_16b_f32 (CB = 2, E = 5..8 only): the input side needs its own A (W_ih) and B (a) operands and its own
// accumulators -- the sides cannot share a chain (the canonical sums keep them apart until the gate formula) -- so a step is 2 x 96
// instructions.  Same function as k_valu_pk.
template <int EP>
__global__ __launch_bounds__(256) void k_mfma4_both(const float *__restrict__ W, const float *__restrict__ h0, int n_off, int T,
                                                    float *__restrict__ out)
{
    static_assert(EP > 4 && EP <= 8, "two column blocks");
    __shared__ __attribute__((aligned(16))) float hv[4][32][8];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    int o = blockIdx.x * 4 + wave;
    const bool valid = o < n_off;
    o = valid ? o : n_off - 1;
    const int b = lane >> 2, i = lane & 3, cb = b & 1, ub = b >> 1;
    float ah[3][32], ai[3][32];
#pragma unroll
    for (int m = 0; m < 3; ++m)
#pragma unroll
        for (int k = 0; k < 32; ++k) {
            ah[m][k] = W[((size_t)o * 96 + 32 * m + 4 * ub + i) * 32 + k];
            ai[m][k] = W[((size_t)o * 96 + 32 * m + 4 * ub + i) * 32 + ((k + 1) & 31)];
        }
    for (int idx = lane; idx < 32 * 8; idx += 64) {
        const int u = idx >> 3, c = idx & 7;
        hv[wave][u][c] = c < EP ? h0[((size_t)o * 32 + u) * EP + c] : 0.0f;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    for (int t = 0; t < T; ++t) {
        float bv[32];
#pragma unroll
        for (int k = 0; k < 32; ++k) bv[k] = hv[wave][k][4 * cb + i];
        f32x4 dh[3], di[3];
#pragma unroll
        for (int m = 0; m < 3; ++m) dh[m] = di[m] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
        for (int k = 0; k < 32; ++k) {
            const float av = 0.5f * bv[k];
#pragma unroll
            for (int m = 0; m < 3; ++m) dh[m] = __builtin_amdgcn_mfma_f32_4x4x1f32(ah[m][k], bv[k], dh[m], 0, 0, 0);
#pragma unroll
            for (int m = 0; m < 3; ++m) di[m] = __builtin_amdgcn_mfma_f32_4x4x1f32(ai[m][k], av, di[m], 0, 0, 0);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
        for (int r = 0; r < 4; ++r)
            hv[wave][4 * ub + r][4 * cb + i] = 0.55f * ((dh[0][r] + dh[1][r]) + dh[2][r]) + 0.05f * ((di[0][r] + di[1][r]) + di[2][r]);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
    if (valid)
        for (int idx = lane; idx < 32 * 8; idx += 64) {
            const int u = idx >> 3, c = idx & 7;
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
    float *o3;
    CK(hipMalloc(&o3, h.size() * 4));
    const float t4 = time_ms([&] { hipLaunchKernelGGL((k_mfma4<EP>), dim3(blocks), dim3(256), 0, 0, dW, dh, n_off, T, o3); });
    // exactness of the 4x4x1 form: ONE step against the k-ascending fmaf chain on the host, bit for bit
    hipLaunchKernelGGL((k_mfma4<EP>), dim3(blocks), dim3(256), 0, 0, dW, dh, n_off, 1, o3);
    CK(hipDeviceSynchronize());
    std::vector<float> r3(h.size());
    CK(hipMemcpy(r3.data(), o3, h.size() * 4, hipMemcpyDeviceToHost));
    size_t bad = 0, checked = 0;
    for (int oo = 0; oo < n_off; oo += 97)
        for (int u = 0; u < 32; ++u)
            for (int e = 0; e < EP; ++e) {
                float gsum[3];
                for (int g = 0; g < 3; ++g) {
                    float acc = 0.0f;
                    for (int k = 0; k < 32; ++k) acc = fmaf(W[((size_t)oo * 96 + 32 * g + u) * 32 + k], h[((size_t)oo * 32 + k) * EP + e], acc);
                    gsum[g] = acc;
                }
                const float want = 0.55f * ((gsum[0] + gsum[1]) + gsum[2]);
                const float got = r3[((size_t)oo * 32 + u) * EP + e];
                unsigned wa, gb;
                memcpy(&wa, &want, 4); memcpy(&gb, &got, 4);
                bad += wa != gb;
                ++checked;
            }
    hipLaunchKernelGGL((k_valu<EP>), dim3(blocks), dim3(256), 0, 0, dW, dh, n_off, 6, o1);
    hipLaunchKernelGGL((k_mfma<EP>), dim3(blocks), dim3(256), 0, 0, dW, dh, n_off, 6, o2);
    CK(hipDeviceSynchronize());
    std::vector<float> r1(h.size()), r2(h.size());
    CK(hipMemcpy(r1.data(), o1, h.size() * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(r2.data(), o2, h.size() * 4, hipMemcpyDeviceToHost));
    double md = 0, mx = 0;
    for (size_t i = 0; i < r1.size(); ++i) { md = fmax(md, fabs((double)r1[i] - r2[i])); mx = fmax(mx, fabs((double)r1[i])); }
    const double steps = (double)n_off * EP * T;
    if constexpr (EP > 4 && EP <= 8) {
        // the production comparison: both sides of the contraction, VALU packed (what ses_gru_lockstep.h runs) against 2 x 96 MFMAs
        const float tp = time_ms([&] { hipLaunchKernelGGL((k_valu_pk<EP>), dim3(blocks), dim3(256), 0, 0, dW, dh, n_off, T, o1); });
        const float tb = time_ms([&] { hipLaunchKernelGGL((k_mfma4_both<EP>), dim3(blocks), dim3(256), 0, 0, dW, dh, n_off, T, o3); });
        hipLaunchKernelGGL((k_valu_pk<EP>), dim3(blocks), dim3(256), 0, 0, dW, dh, n_off, 6, o1);
        hipLaunchKernelGGL((k_mfma4_both<EP>), dim3(blocks), dim3(256), 0, 0, dW, dh, n_off, 6, o3);
        CK(hipDeviceSynchronize());
        std::vector<float> p1(h.size()), p3(h.size());
        CK(hipMemcpy(p1.data(), o1, h.size() * 4, hipMemcpyDeviceToHost));
        CK(hipMemcpy(p3.data(), o3, h.size() * 4, hipMemcpyDeviceToHost));
        double pd = 0;
        for (size_t q = 0; q < p1.size(); ++q) pd = fmax(pd, fabs((double)p1[q] - p3[q]));
        printf("E = %2d, %d offspring, BOTH sides (input + hidden, the production contraction): VALU v_pk_fma_f32 %8.3f ms   MFMA 4x4x1_16b (2 x 96 per step) "
               "%8.3f ms   MFMA/VALU time = %.2f   max|diff| = %.2e\n", EP, n_off, tp, tb, tb / tp, pd);
    }
    printf("E = %2d, %d offspring, %d steps: VALU %8.3f ms (%6.2f ns/env-step)   MFMA 16x16x4 %8.3f ms (%6.2f ns/env-step)   "
           "MFMA/VALU time = %.2f   max|diff| = %.2e (max|h| %.2e)   MFMA 4x4x1_16b %8.3f ms (%6.2f ns/env-step)  4x4x1/VALU time = %.2f   "
           "4x4x1 one step vs k-ascending fmaf chain: %zu of %zu differ\n",
           EP, n_off, T, tv, tv * 1e6 / steps, tm, tm * 1e6 / steps, tm / tv, md, mx, t4, t4 * 1e6 / steps, t4 / tv, bad, checked);
    hipFree(dW); hipFree(dh); hipFree(o1); hipFree(o2); hipFree(o3);
}

int main()
{
    for (int n_off : {4096, 16384}) {
        run<4>(n_off, 200);
        run<5>(n_off, 200);
        run<8>(n_off, 200);
        run<16>(n_off, 200);
    }
    return 0;
}
