// ses_gru_mfma4.h -- GRU policy step for up to 8 episodes of one offspring on the matrix cores, in 4x4x1 blocks (round 6).
//
// v_mfma_f32_4x4x1_16b_f32 is 16 INDEPENDENT blocks D_b[4x4] += A_b[4x1] . B_b[1x4]: every block has its own A and its own
// B, which is what "per-offspring weights, a handful of episodes" needs -- the columns are padded to a multiple of FOUR
// episodes, not of sixteen (ses_gru_mfma.h's 16x16x4 tile), and one instruction advances ONE k, so a run of them IS the
// k-ascending fma chain, bit for bit (tools/mfma_vs_valu_gru.hip: 0 of 86 528 outputs differ from fmaf on the host).
//
// Lane roles (lane l): block b = l >> 2, i = l & 3; episode block cb = b & 1, unit block ub = b >> 1.
//   A operand of the lane = row (4 ub + i) of a weight matrix at column k        (W_hh: 96 values per lane in VGPRs for the rollout;
//                                                                                 W_ih: read from the wave's LDS block, four k per read)
//   B operand             = activation k of episode e = 4 cb + i                 (k-pairs of (a_k, h_k) from LDS)
//   D register r          = (unit 4 ub + r, episode e)
// so a lane ends a step with the gate pre-activations of FOUR units of ONE episode: the three gates of a (unit, episode) and
// its hidden state are in one lane, no half-sum exchange (the VALU lockstep form spends 6 v_permlane32_swap per episode pair).
// Canonical sums as everywhere (ses_gru.h): gate row = (bias + chain k < 16) + (chain k >= 16), input and hidden side apart;
// fc1 = bias-first chain over the S inputs (S MFMAs with C = b1); fc2 and the env as in the lockstep form.
// Cost per step: 192 + S MFMAs of 8 cycles for ANY number of episodes up to 8 -- against 48 v_pk_fma_f32 per episode in the
// VALU form.  Measured, POMDP CartPole, 4096 offspring x 500 steps (profiles/r06_time_gru.txt): 3.05 ms for every E <= 8; the VALU
// lockstep kernel takes 2.45 / 2.70 / 3.27 / 3.52 ms at 5 / 6 / 7 / 8 episodes -- ses_rollout takes this step from 7 (knob
// "gru_mfma4_min_e").  History: with all 192 gate weights of a lane in registers (the two column blocks of an offspring need the same
// A rows: 2 x duplication) the kernel needed 360 registers = ONE wave per SIMD and took 4.01 ms -- a lone wave runs the ~480 non-MFMA
// instructions of a step with every stall exposed; W_ih moved to LDS: 250 VGPRs, two waves per SIMD, 3.05 ms.
#pragma once
#include <hip/hip_runtime.h>

#include <cstddef>

#include "ses_gru.h"

namespace ses {

constexpr int G4_EB = 8;          // episodes per batch = 2 column blocks of 4
constexpr int G4_WROW = 36;       // floats per W_ih row in LDS
constexpr int G4_AH = 34;         // (a, h) pairs per episode row: 32 + 2 of padding -> rows 272 B apart, the 8 rows' ds_read_b128 hit 8 bank groups

typedef float g4_f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ g4_f32x4 mfma_4x4x1(float a, float b, g4_f32x4 c)
{
    return __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c, 0, 0, 0);
}

template <int S, int A>
struct alignas(16) GruMfma4Lds {
    float ah[G4_EB][G4_AH][2];    // [episode][k][0] fc1 activation a_k, [1] hidden state h_k
    float y[G4_EB][36];           // tanh(h') for fc2 (144-B rows, as in the lockstep form)
    float obs[G4_EB][8];          // masked observations (S <= 8)
    float w2[A][32];
    float bi[96], bh[96];         // gate biases, row = 32 gate + unit (read as float4 accumulator fragments: 16-byte aligned)
    float b1[32];
    float wih[96][G4_WROW];       // W_ih, row = 32 gate + unit, 36 floats per row (144 B: the rows a wave reads together -- 32 of them,
                                  // four k at a time -- fall on different bank groups); W_hh stays in registers
    float b2[A];                  // last: its 4 A bytes must not push a float4-read array off its alignment
};
typedef GruMfma4Lds<4, 2> G4LdsCartPole;
static_assert(offsetof(G4LdsCartPole, bi) % 16 == 0 && offsetof(G4LdsCartPole, bh) % 16 == 0 && offsetof(G4LdsCartPole, b1) % 16 == 0 &&
                  offsetof(G4LdsCartPole, wih) % 16 == 0 &&
                  offsetof(G4LdsCartPole, obs) % 16 == 0 && offsetof(G4LdsCartPole, y) % 16 == 0 && sizeof(G4LdsCartPole) % 16 == 0,
              "every array that is read with ds_read_b128 stays 16-byte aligned in every wave's copy");

template <int S, int A>
struct GruMfma4 {
    static_assert(S == 4 || S == 8, "observations arrive as one or two float4");
    float w1[S];                  // W1[4 ub + i][s]
    float whh[3][32];             // W_hh[32 g + 4 ub + i][k]: resident (W_ih is read from the wave's LDS block, see step())

    __device__ __forceinline__ void load(const float *__restrict__ theta, int lane, GruMfma4Lds<S, A> &lds)
    {
        const int b = lane >> 2, i = lane & 3, ub = b >> 1, row = 4 * ub + i;
        const float *p = theta;
#pragma unroll
        for (int s = 0; s < S; ++s) w1[s] = p[row * S + s];
        p += H * S;
        if (lane < 32) lds.b1[lane] = p[lane];
        p += H;
        const float *pih = p, *phh = p + 3 * H * H, *pbi = p + 6 * H * H, *pbh = pbi + 3 * H;
#pragma unroll
        for (int g = 0; g < 3; ++g)
#pragma unroll
            for (int k = 0; k < 32; ++k) whh[g][k] = phh[(g * H + row) * H + k];
        for (int q = lane; q < 96 * 32; q += 64) lds.wih[q >> 5][q & 31] = pih[q];      // coalesced: 3072 consecutive floats
        for (int q = lane; q < 96; q += 64) {
            lds.bi[q] = pbi[q];
            lds.bh[q] = pbh[q];
        }
        p = pbh + 3 * H;
        if (lane < 32) {
#pragma unroll
            for (int o = 0; o < A; ++o) lds.w2[o][lane] = p[o * H + lane];
        }
        if (lane < A) lds.b2[lane] = p[A * H + lane];
    }

    __device__ static __forceinline__ g4_f32x4 frag(const float *v, int ub)      // v[4 ub .. 4 ub + 3] as an accumulator
    {
        const float4 t = reinterpret_cast<const float4 *>(v)[ub];
        return g4_f32x4{t.x, t.y, t.z, t.w};
    }

    // One time step for the 8 episode columns.  lds.obs must hold the (masked) observations; hreg[r] is this lane's copy of
    // h[unit 4 ub + r][episode e].  On return lds.y holds tanh(h') and lds.ah[.][.][1] the new hidden state.
    __device__ __forceinline__ void step(const TanhEntry *tab, GruMfma4Lds<S, A> &lds, float (&hreg)[4], int lane) const
    {
        const int b = lane >> 2, i = lane & 3, cb = b & 1, ub = b >> 1, e = 4 * cb + i;
        // ---- fc1: C = b1, one MFMA per input
        {
            g4_f32x4 d = frag(lds.b1, ub);
            const float4 o0 = *reinterpret_cast<const float4 *>(&lds.obs[e][0]);
            d = mfma_4x4x1(w1[0], o0.x, d);
            d = mfma_4x4x1(w1[1], o0.y, d);
            d = mfma_4x4x1(w1[2], o0.z, d);
            d = mfma_4x4x1(w1[3], o0.w, d);
            if constexpr (S == 8) {
                const float4 o1 = *reinterpret_cast<const float4 *>(&lds.obs[e][4]);
                d = mfma_4x4x1(w1[4], o1.x, d);
                d = mfma_4x4x1(w1[5], o1.y, d);
                d = mfma_4x4x1(w1[6], o1.z, d);
                d = mfma_4x4x1(w1[7], o1.w, d);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) lds.ah[e][4 * ub + r][0] = tanh_(tab, d[r]);
        }
        wave_lds_sync();
        // ---- gate contractions: 12 accumulation chains (3 gates x input / hidden x lower / upper half of k), interleaved
        g4_f32x4 il[3], iu[3], hl[3], hu[3];
#pragma unroll
        for (int g = 0; g < 3; ++g) {
            il[g] = frag(lds.bi + 32 * g, ub);
            hl[g] = frag(lds.bh + 32 * g, ub);
            iu[g] = g4_f32x4{0.0f, 0.0f, 0.0f, 0.0f};
            hu[g] = g4_f32x4{0.0f, 0.0f, 0.0f, 0.0f};
        }
        const float4 *row = reinterpret_cast<const float4 *>(&lds.ah[e][0][0]);     // (a_k, h_k, a_k+1, h_k+1) per read
        const float4 *wrow[3];                                                      // this lane's W_ih rows, four k per read
#pragma unroll
        for (int g = 0; g < 3; ++g) wrow[g] = reinterpret_cast<const float4 *>(&lds.wih[32 * g + 4 * ub + i][0]);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            // k = 4 q .. 4 q + 3 (lower half) and 16 + 4 q .. 19 + 4 q (upper half)
            const float4 lo0 = row[2 * q], lo1 = row[2 * q + 1], up0 = row[8 + 2 * q], up1 = row[9 + 2 * q];
            const float xa_lo[4] = {lo0.x, lo0.z, lo1.x, lo1.z}, xh_lo[4] = {lo0.y, lo0.w, lo1.y, lo1.w};
            const float xa_up[4] = {up0.x, up0.z, up1.x, up1.z}, xh_up[4] = {up0.y, up0.w, up1.y, up1.w};
            float4 wl[3], wu[3];
#pragma unroll
            for (int g = 0; g < 3; ++g) {
                wl[g] = wrow[g][q];
                wu[g] = wrow[g][4 + q];
            }
#pragma unroll
            for (int c = 0; c < 4; ++c) {
#pragma unroll
                for (int g = 0; g < 3; ++g) {
                    const float wlc = c == 0 ? wl[g].x : (c == 1 ? wl[g].y : (c == 2 ? wl[g].z : wl[g].w));
                    const float wuc = c == 0 ? wu[g].x : (c == 1 ? wu[g].y : (c == 2 ? wu[g].z : wu[g].w));
                    il[g] = mfma_4x4x1(wlc, xa_lo[c], il[g]);
                    hl[g] = mfma_4x4x1(whh[g][4 * q + c], xh_lo[c], hl[g]);
                    iu[g] = mfma_4x4x1(wuc, xa_up[c], iu[g]);
                    hu[g] = mfma_4x4x1(whh[g][16 + 4 * q + c], xh_up[c], hu[g]);
                }
            }
        }
        wave_lds_sync();                                  // every lane holds what it needs of the old a / h rows
        float hn[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float tir = il[0][r] + iu[0][r], thr = hl[0][r] + hu[0][r];
            const float tiz = il[1][r] + iu[1][r], thz = hl[1][r] + hu[1][r];
            const float tin = il[2][r] + iu[2][r], thn = hl[2][r] + hu[2][r];
            const float pr = 0.5f * (tir + thr), pz = 0.5f * (tiz + thz);
            float ur, uz;
            const int32_t ir = tanh_index(pr, ur), iz = tanh_index(pz, uz);
            const TanhEntry er = tab[ir], ez = tab[iz];
            const float rg = fma_(0.5f, tanh_eval(er, ur, pr), 0.5f);
            const float zg = fma_(0.5f, tanh_eval(ez, uz, pz), 0.5f);
            const float ng = tanh_(tab, fma_(rg, thn, tin));
            hn[r] = fma_(zg, hreg[r] - ng, ng);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            hreg[r] = hn[r];
            lds.ah[e][4 * ub + r][1] = hn[r];
            lds.y[e][4 * ub + r] = tanh_(tab, hn[r]);
        }
        wave_lds_sync();
    }

    // fc2 of episode (lane & 7), spread over the 8 lanes that own it: exactly GruLockstep::logits_of (ses_gru_lockstep.h)
    __device__ static __forceinline__ void logits_of(const GruMfma4Lds<S, A> &lds, int lane, float (&logits)[A])
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
