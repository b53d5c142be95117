// ses_gru_lockstep.h -- GRU policy rollout with all episodes of an offspring advancing in lockstep.
//
// One offspring per wavefront as in ses_gru.h (lane = hidden unit j x k-half kh, 6 x 16 gate weights per
// lane in VGPRs), but instead of playing the E episodes one after the other, up to EB = 8 of them advance
// together each time step.  That removes the replication of everything that is not the gate contraction:
//   - physics / reward / termination: lane l OWNS episode (l & 7) -- one pass of the env code serves all
//     episodes (the sequential form ran it once per episode on 64 identical lanes);
//   - gate non-linearities and fc1: the lower half finishes the even episodes, the upper half the odd ones
//     (the two half-sums of an episode pair cross over with ONE v_permlane32_swap per gate row);
//   - fc2: the owner lane of an episode reads the 32 tanh(h') values from LDS and evaluates the canonical
//     chain-of-4 + tree in registers (W2 lives in LDS, broadcast reads).
// Arithmetic and its order are exactly those of ses_gru.h / oracle/ses_oracle.c, so returns are unchanged.
// Measured (POMDP CartPole, 4096 offspring x 5 episodes x 500 steps): sequential 4.2 ms -> see DESIGN.md.
#pragma once
#include <hip/hip_runtime.h>

#include "ses_gru.h"

namespace ses {

constexpr int GL_EB = 8;   // episodes per lockstep batch = owner slots per 8-lane group

// wave-private LDS block
// alignas(16): every row that is read with ds_read_b128 must stay 16-byte aligned in EVERY wave's copy; with a
// size that is not a multiple of 16 the odd waves' copies were 8-byte aligned and each b128 read cost ~27 LDS
// cycles instead of 4 (SQ_LDS_IDX_ACTIVE), making the kernel LDS-bound.
template <int S, int A>
struct alignas(16) GruLockstepLds {
    float a[GL_EB][32];      // fc1 activations
    float h[GL_EB][32];      // hidden state
    float y[GL_EB][36];      // tanh(h') for fc2; 144-B rows: the 8 owner rows fall on 8 different bank groups
    float obs[GL_EB][8];     // observations (S <= 8), masked
    float w2[A][32];
    float b2[A];
};

// x = (lower: value for the upper half, upper: value for the lower half) -> exchanged halves
__device__ __forceinline__ float swap_halves(float x, int kh)
{
    float a = x, b = x;
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
    // after the swap: a.upper = x.lower, b.lower = x.upper
    return kh ? a : b;
}

template <int S, int A>
struct GruLockstep {
    float w1[S], b1;
    float wih[3][16], whh[3][16];
    float bih[3], bhh[3];

    __device__ __forceinline__ void load(const float *__restrict__ theta, int lane, GruLockstepLds<S, A> &lds)
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
            for (int k = 0; k < 16; ++k) {
                wih[g][k] = pih[(g * H + j) * H + 16 * kh + k];
                whh[g][k] = phh[(g * H + j) * H + 16 * kh + k];
            }
            bih[g] = kh ? 0.0f : pbi[g * H + j];
            bhh[g] = kh ? 0.0f : pbh[g * H + j];
        }
        p = pbh + 3 * H;
        if (kh == 0) {
#pragma unroll
            for (int o = 0; o < A; ++o) lds.w2[o][j] = p[o * H + j];
        }
        if (lane < A) lds.b2[lane] = p[A * H + lane];
    }

    // One time step for the episodes [0, 2*NP) of the batch.  hreg[p] is this lane's hidden unit for the episode
    // it finishes in pair p (episode 2p + kh).  lds.obs must hold the (masked) observations; on return lds.y holds
    // tanh(h') and lds.h the new hidden state.
    // ODD: the last pair holds a single real episode; its partner's contraction is skipped (its partial sums are
    // taken as 0, the upper half then finishes a dummy episode whose rows nobody reads).
    template <int NP, bool ODD>
    __device__ __forceinline__ void step(const TanhEntry *tab, GruLockstepLds<S, A> &lds, float (&hreg)[NP], int lane) const
    {
        const int j = lane & 31, kh = lane >> 5;
        // fc1 for my episodes
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            const int me = 2 * p + kh;
            float acc = b1;
#pragma unroll
            for (int k = 0; k < S; ++k) acc = fma_(w1[k], lds.obs[me][k], acc);
            lds.a[me][j] = tanh_(tab, acc);
        }
        wave_lds_sync();
        float hn[NP];
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            float part[2][6];
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int e = 2 * p + q;
                if (ODD && p == NP - 1 && q == 1) {
#pragma unroll
                    for (int g = 0; g < 6; ++g) part[q][g] = 0.0f;
                    continue;
                }
                // all 8 slice reads of this episode are issued before the first fma needs one of them
                const float4 *va = reinterpret_cast<const float4 *>(&lds.a[e][16 * kh]);
                const float4 *vh = reinterpret_cast<const float4 *>(&lds.h[e][16 * kh]);
                float4 xa[4], xh[4];
#pragma unroll
                for (int c4 = 0; c4 < 4; ++c4) { xa[c4] = va[c4]; xh[c4] = vh[c4]; }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int g = 0; g < 3; ++g) { part[q][g] = bih[g]; part[q][3 + g] = bhh[g]; }
#pragma unroll
                for (int c4 = 0; c4 < 4; ++c4) {
                    const float ea[4] = {xa[c4].x, xa[c4].y, xa[c4].z, xa[c4].w};
                    const float eh[4] = {xh[c4].x, xh[c4].y, xh[c4].z, xh[c4].w};
#pragma unroll
                    for (int c = 0; c < 4; ++c)
#pragma unroll
                        for (int g = 0; g < 3; ++g) {
                            part[q][g] = fma_(wih[g][4 * c4 + c], ea[c], part[q][g]);
                            part[q][3 + g] = fma_(whh[g][4 * c4 + c], eh[c], part[q][3 + g]);
                        }
                }
            }
            // I finish episode 2p + kh: keep my partial of it, hand over my partial of the other one
            float tot[6];
#pragma unroll
            for (int g = 0; g < 6; ++g) {
                const float keep = kh ? part[1][g] : part[0][g];
                const float give = kh ? part[0][g] : part[1][g];
                tot[g] = keep + swap_halves(give, kh);   // lo + hi (lower half) / hi + lo (upper half): same bits
            }
            // r and z table lookups issued together
            const float pr = 0.5f * (tot[0] + tot[3]), pz = 0.5f * (tot[1] + tot[4]);
            float ur, uz;
            const int32_t ir = tanh_index(pr, ur), iz = tanh_index(pz, uz);
            const TanhEntry er = tab[ir], ez = tab[iz];
            const float r = fma_(0.5f, tanh_eval(er, ur, pr), 0.5f);
            const float z = fma_(0.5f, tanh_eval(ez, uz, pz), 0.5f);
            const float n = tanh_(tab, fma_(r, tot[5], tot[2]));
            hn[p] = fma_(z, hreg[p] - n, n);
        }
        wave_lds_sync();                                  // every lane has consumed the old h rows
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            const int me = 2 * p + kh;
            hreg[p] = hn[p];
            lds.h[me][j] = hn[p];
            lds.y[me][j] = tanh_(tab, hn[p]);
        }
        wave_lds_sync();
    }

    // fc2 of episode e on its owner lane: canonical chain of 4 + tree over the 8 groups + bias
    __device__ __forceinline__ void logits_of(const GruLockstepLds<S, A> &lds, int e, float (&logits)[A]) const
    {
        // group by group: one quad of y and one quad of every W2 row at a time (no 32-value register array)
        float pg[A][8];
        const float4 *vy = reinterpret_cast<const float4 *>(&lds.y[e][0]);
#pragma unroll
        for (int g = 0; g < 8; ++g) {
            const float4 y = vy[g];
#pragma unroll
            for (int o = 0; o < A; ++o) {
                const float4 w = reinterpret_cast<const float4 *>(&lds.w2[o][0])[g];
                float acc = w.x * y.x;
                acc = fma_(w.y, y.y, acc);
                acc = fma_(w.z, y.z, acc);
                acc = fma_(w.w, y.w, acc);
                pg[o][g] = acc;
            }
        }
#pragma unroll
        for (int o = 0; o < A; ++o) {
            logits[o] = (((pg[o][0] + pg[o][1]) + (pg[o][2] + pg[o][3])) + ((pg[o][4] + pg[o][5]) + (pg[o][6] + pg[o][7]))) + lds.b2[o];
        }
    }
};

}  // namespace ses
