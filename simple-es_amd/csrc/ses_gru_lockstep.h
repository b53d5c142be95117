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
    float ah[GL_EB][32][2];  // [k][0] fc1 activation a_k, [k][1] hidden state h_k: one b128 read = two (a, h) pairs
    float y[GL_EB][36];      // tanh(h') for fc2; 144-B rows: the 8 owner rows fall on 8 different bank groups
    float obs[GL_EB][8];     // observations (S <= 8), masked
    float w2[A][32];
    float b2[A];
};

// p0 / p1: this lane's partial sums (its k-half) for the even / odd episode of a pair.  The lower half of the wave
// finishes the even episode, the upper half the odd one.  v_permlane32_swap exchanges p0's upper half with p1's
// lower half; afterwards p0 = (own partial | other half's partial) and p1 = (other | own), so p0 + p1 is
// lower-k + upper-k of the episode each lane finishes: one swap and one add, no selects.
__device__ __forceinline__ float pair_total(float p0, float p1)
{
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(p0), "+v"(p1));
    return p0 + p1;
}

// (input-side, hidden-side) halves of one gate row's contraction ride one v_pk_fma_f32: the weights
// (W_ih[g][j][k], W_hh[g][j][k]) sit in a register pair, the operands (a_k, h_k) arrive as a pair from LDS, the two
// partial sums are the pair's halves.  96 -> 48 contraction instructions per episode; each half is an ordinary
// IEEE fma in the same k order, so the sums keep their bits.
typedef float gl_v2f __attribute__((ext_vector_type(2)));

template <int S, int A>
struct GruLockstep {
    float w1[S], b1;
    gl_v2f w[3][16];         // {W_ih, W_hh}[g][j][16 kh + k]
    gl_v2f b[3];             // {b_ih, b_hh}[g][j] on the lower half, 0 on the upper half

    // WITH_LDS = false: the register part only (kernels that stream the weights of several offspring through one wave
    // reload it every step and write W2 / b2 to LDS once)
    template <bool WITH_LDS = true>
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
                w[g][k] = gl_v2f{pih[(g * H + j) * H + 16 * kh + k], phh[(g * H + j) * H + 16 * kh + k]};
            }
            b[g] = kh ? gl_v2f{0.0f, 0.0f} : gl_v2f{pbi[g * H + j], pbh[g * H + j]};
        }
        if constexpr (WITH_LDS) {
            p = pbh + 3 * H;
            if (kh == 0) {
#pragma unroll
                for (int o = 0; o < A; ++o) lds.w2[o][j] = p[o * H + j];
            }
            if (lane < A) lds.b2[lane] = p[A * H + lane];
        }
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
            lds.ah[me][j][0] = tanh_(tab, acc);
        }
        wave_lds_sync();
        float hn[NP];
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            gl_v2f part[2][3];                             // [q][g] = {input-side, hidden-side} partial sums
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int e = 2 * p + q;
                if (ODD && p == NP - 1 && q == 1) {
#pragma unroll
                    for (int g = 0; g < 3; ++g) part[q][g] = gl_v2f{0.0f, 0.0f};
                    continue;
                }
                // all 8 slice reads of this episode are issued before the first fma needs one of them
                const float4 *vx = reinterpret_cast<const float4 *>(&lds.ah[e][16 * kh][0]);
                float4 x[8];
#pragma unroll
                for (int c2 = 0; c2 < 8; ++c2) x[c2] = vx[c2];
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int g = 0; g < 3; ++g) part[q][g] = b[g];
#pragma unroll
                for (int c2 = 0; c2 < 8; ++c2) {
                    const gl_v2f x0 = {x[c2].x, x[c2].y}, x1 = {x[c2].z, x[c2].w};   // (a, h) at k = 2 c2, 2 c2 + 1
#pragma unroll
                    for (int g = 0; g < 3; ++g) part[q][g] = __builtin_elementwise_fma(w[g][2 * c2], x0, part[q][g]);
#pragma unroll
                    for (int g = 0; g < 3; ++g) part[q][g] = __builtin_elementwise_fma(w[g][2 * c2 + 1], x1, part[q][g]);
                }
            }
            // I finish episode 2p + kh: keep my partial of it, hand over my partial of the other one
            float tot[6];                                  // 0..2 input side (r, z, n), 3..5 hidden side
#pragma unroll
            for (int g = 0; g < 3; ++g) {
#pragma unroll
                for (int s = 0; s < 2; ++s)
                    tot[3 * s + g] = pair_total(s ? part[0][g].y : part[0][g].x, s ? part[1][g].y : part[1][g].x);
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
            lds.ah[me][j][1] = hn[p];
            lds.y[me][j] = tanh_(tab, hn[p]);
        }
        wave_lds_sync();
    }

    // fc2 of episode (lane & 7), spread over the 8 lanes that own it: lane l evaluates group (l >> 3) -- the
    // canonical in-order chain over 4 consecutive hidden units -- and the balanced tree over the 8 groups is three
    // exchange-and-add levels (lane bits 3, 4, 5): a DPP rotate inside the 16-lane row, then v_permlane16_swap and
    // v_permlane32_swap on two copies, which leave (even | even) and (odd | odd) so that their sum is the pair
    // total in every lane, even operand first.  Every lane of the slot ends with the same logits.
    __device__ __forceinline__ void logits_of(const GruLockstepLds<S, A> &lds, int lane, float (&logits)[A]) const
    {
        const int e = lane & 7, grp = lane >> 3;
        const float4 y = reinterpret_cast<const float4 *>(&lds.y[e][0])[grp];
#pragma unroll
        for (int o = 0; o < A; ++o) {
            const float4 w = reinterpret_cast<const float4 *>(&lds.w2[o][0])[grp];
            float acc = w.x * y.x;
            acc = fma_(w.y, y.y, acc);
            acc = fma_(w.z, y.z, acc);
            acc = fma_(w.w, y.w, acc);
            acc = acc + dpp_mov<DPP_ROW_ROR8>(acc);                        // groups (0,1) (2,3) (4,5) (6,7)
            float lo = acc, hi = acc;
            asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 1" : "+v"(lo), "+v"(hi));
            acc = lo + hi;                                                  // (01)+(23), (45)+(67)
            lo = acc;
            hi = acc;
            asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(lo), "+v"(hi));
            logits[o] = (lo + hi) + lds.b2[o];
        }
    }
};

}  // namespace ses
