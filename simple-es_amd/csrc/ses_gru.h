// ses_gru.h -- GymEnvModel forward with the nn.GRU(32,32) cell (networks/neural_network.py:20-36), one
// offspring per wavefront.
//
// Lane l of the wave owns hidden unit j = l & 31 and the k-half kh = l >> 5 of every 32-long gate row:
// it keeps 6 x 16 GRU weights (+ its fc1 row and fc2 column) in VGPRs for the whole rollout, so the
// 27 KB weight set of an offspring is read from HBM exactly once.  The activation vectors a[32] (fc1
// output) and h[32] (hidden state) are exchanged through 256 bytes of wave-private LDS, interleaved as
// (a_k, h_k) pairs: each half reads its 16 pairs with eight broadcast ds_read_b128 and advances the
// input-side and hidden-side sums of a gate row with one v_pk_fma_f32 per k (round 6; 97 offspring x 5
// episodes as lone waves: 0.374 -> 0.350 ms per rollout, 4096 offspring sequential 4.08 -> 3.99 ms).  The two half-sums of a gate row are combined
// with v_permlane32_swap.  fc2 runs across the lanes with DPP row shifts.
//
// Why not MFMA: the gate contraction is [96x32] x [32 x E] per offspring with E = 5 episodes; a
// v_mfma_f32_32x32x2_f32 tile would be 5/32 occupied, and fp32 MFMA has the same FLOP rate as the VALU on
// gfx950, so the padded tile would be ~3x slower than the VALU form below.
//
// Canonical arithmetic (restated by oracle/ses_oracle.c):
//   gate row : (bias + fma chain over k = 0..15) + (0 + fma chain over k = 16..31)
//   r = sigmoid(gi_r + gh_r); z = sigmoid(gi_z + gh_z); n = tanh(fma(r, gh_n, gi_n)); h' = fma(z, h - n, n)
//   fc2      : chain of 4 consecutive units, balanced tree over the 8 groups, + bias (as in ses_policy.h)
#pragma once
#include <hip/hip_runtime.h>

#include "ses_math.h"
#include "ses_policy.h"

namespace ses {

constexpr int DPP_ROW_SHR1 = 0x111;
constexpr int DPP_ROW_SHR4 = 0x114;
constexpr int DPP_ROW_SHR8 = 0x118;
constexpr int DPP_ROW_BCAST15 = 0x142;

// compiler + hardware ordering point for wave-private LDS traffic (DS ops of one wave execute in order;
// this only stops the compiler from moving LDS accesses across it)
__device__ __forceinline__ void wave_lds_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// x holds a half-sum: lanes 0..31 the k<16 part, lanes 32..63 the k>=16 part of the same 32 units.
// Returns lo + hi on every lane (identical bits in both halves).
__device__ __forceinline__ float half_pair_sum(float x)
{
    float a = x, b = x;
    // after the swap: a = (lo, lo), b = (hi, hi)
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
    return a + b;
}

template <int S, int A>
struct GruSlice {
    float w1[S], b1;
    // round 6: (W_ih, W_hh) of a gate row ride one register pair, like ses_gru_lockstep.h: the input-side and the hidden-side
    // sums advance with ONE v_pk_fma_f32 per k (each half an ordinary IEEE fma in the same k order) -- 48 instead of 96
    // contraction instructions per step.  The wave-private LDS vector is interleaved to match: vec[2 k] = a_k, vec[2 k + 1] = h_k.
    typedef float v2f __attribute__((ext_vector_type(2)));
    v2f w[3][16];           // {W_ih, W_hh}[g][j][16 kh + k]
    v2f b[3];               // {b_ih, b_hh}[g][j]; zero on the k>=16 half (the bias belongs to the first half-sum)
    float w2[A], b2[A];

    __device__ __forceinline__ void load(const float *__restrict__ theta, int lane)
    {
        const int j = lane & 31, kh = lane >> 5;
        const float *p = theta;
#pragma unroll
        for (int k = 0; k < S; ++k) w1[k] = p[j * S + k];
        p += H * S;
        b1 = p[j];
        p += H;
        const float *pih = p, *phh = p + 3 * H * H, *pbi = p + 6 * H * H, *pbh = pbi + 3 * H;
#pragma unroll
        for (int g = 0; g < 3; ++g) {
#pragma unroll
            for (int k = 0; k < 16; ++k) w[g][k] = v2f{pih[(g * H + j) * H + 16 * kh + k], phh[(g * H + j) * H + 16 * kh + k]};
            b[g] = kh ? v2f{0.0f, 0.0f} : v2f{pbi[g * H + j], pbh[g * H + j]};
        }
        p = pbh + 3 * H;
#pragma unroll
        for (int a = 0; a < A; ++a) {
            w2[a] = p[a * H + j];
            b2[a] = p[A * H + a];
        }
    }

    // One forward pass.  obs is wave-uniform, h is this lane's hidden unit (updated in place),
    // vec = 64 floats of wave-private LDS (vec[2 k] = a_k, vec[2 k + 1] = h_k; the h entries must already hold the CURRENT state).
    __device__ __forceinline__ void forward(const TanhEntry *tab, const float (&obs)[S], float &h, float *vec,
                                            int lane, float (&logits)[A]) const
    {
        const int j = lane & 31, kh = lane >> 5;
        float acc = b1;
#pragma unroll
        for (int k = 0; k < S; ++k) acc = fma_(w1[k], obs[k], acc);
        const float a = tanh_(tab, acc);
        if (kh == 0) vec[2 * j] = a;
        wave_lds_sync();
        v2f part[3];
#pragma unroll
        for (int g = 0; g < 3; ++g) part[g] = b[g];
        const float4 *vx = reinterpret_cast<const float4 *>(vec + 32 * kh);          // this half's 16 (a, h) pairs
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const float4 x = vx[q];
            const v2f x0 = {x.x, x.y}, x1 = {x.z, x.w};                               // (a, h) at k = 2 q, 2 q + 1
#pragma unroll
            for (int g = 0; g < 3; ++g) part[g] = __builtin_elementwise_fma(w[g][2 * q], x0, part[g]);
#pragma unroll
            for (int g = 0; g < 3; ++g) part[g] = __builtin_elementwise_fma(w[g][2 * q + 1], x1, part[g]);
        }
        float gi[3], gh[3];
#pragma unroll
        for (int g = 0; g < 3; ++g) { gi[g] = half_pair_sum(part[g].x); gh[g] = half_pair_sum(part[g].y); }
        const float r = sigmoid_(tab, gi[0] + gh[0]);
        const float z = sigmoid_(tab, gi[1] + gh[1]);
        const float n = tanh_(tab, fma_(r, gh[2], gi[2]));
        const float hn = fma_(z, h - n, n);
        h = hn;
        wave_lds_sync();                 // every lane has consumed the old h slice
        if (kh == 0) vec[2 * j + 1] = hn;
        const float y = tanh_(tab, hn);  // neural_network.py:27
#pragma unroll
        for (int o = 0; o < A; ++o) {
            // chain of 4 consecutive units: valid on lanes = 3 (mod 4)
            float t = w2[o] * y;
            t = fma_(w2[o], y, dpp_mov<DPP_ROW_SHR1>(t));
            t = fma_(w2[o], y, dpp_mov<DPP_ROW_SHR1>(t));
            t = fma_(w2[o], y, dpp_mov<DPP_ROW_SHR1>(t));
            // tree over the 8 groups (lanes 3, 7, ..., 31): valid on lane 31 (and 63)
            t = t + dpp_mov<DPP_ROW_SHR4>(t);
            t = t + dpp_mov<DPP_ROW_SHR8>(t);
            t = t + dpp_mov<DPP_ROW_BCAST15>(t);
            logits[o] = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, t), 31)) + b2[o];
        }
    }
};

}  // namespace ses
