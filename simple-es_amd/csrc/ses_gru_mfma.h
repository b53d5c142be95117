// ses_gru_mfma.h -- GRU policy step for up to 16 episodes of one offspring on the matrix cores.
//
// With the reference's default of 5 episodes per offspring the gate contraction [96 x 32] . [32 x E] fills 5 of
// the 16 columns of a v_mfma_f32_16x16x4_f32 tile and the VALU form (ses_gru_lockstep.h) is faster.  The MFMA
// form costs the same for any E <= 16 and overtakes the VALU form at E = 12 (5.1 ms against 5.6 ms; 7.2 ms at
// E = 16), so ses_rollout switches to this step for eval_ep_num >= 12.
//
// Exactness.  v_mfma_f32_16x16x4_f32 accumulates its 4 products one after the other with one rounding each: a
// run of MFMAs over ascending k-blocks IS the k-ascending fmaf chain, bit for bit (tools/mfma_exact.hip, 512 000
// outputs).  The canonical sums are therefore kept as they are:
//   fc1      : C = b1,            K = S ascending                        (bias-first fma chain)
//   gate row : (C = bias, k-blocks 0..3) + (C = 0, k-blocks 4..7), input side and hidden side separately
//   fc2      : one MFMA per group of 4 hidden units with C = 0 (first product plain), balanced tree, + bias
// Non-linearities, the GRU update and the env run on the VALU exactly as in the lockstep form.
//
// Fragment layouts (lane l: li = l & 15, lg = l >> 4):  A[i = li][k = lg],  B[k = lg][n = li],
// C/D register r = element [row 4 lg + r][column li].  Columns are episodes; rows are hidden units / gate rows
// / outputs.  Activations live in LDS k-major ([k][episode]), so the B fragment of k-block kb is the 64
// consecutive floats starting at 64 kb: one conflict-free ds_read_b32 per fragment.  Weights are A fragments held
// in VGPRs for the whole rollout (96 for the gates, like the VALU forms).
#pragma once
#include <hip/hip_runtime.h>

#include "ses_gru.h"

namespace ses {

constexpr int GM_EB = 16;   // episodes per batch = tile columns

typedef float gm_f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ gm_f32x4 mfma4(float a, float b, gm_f32x4 c)
{
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

// wave-private LDS block
template <int S, int A>
struct alignas(16) GruMfmaLds {
    float aT[32][GM_EB];     // fc1 activations, [unit][episode]
    float hT[32][GM_EB];     // hidden state
    float yT[32][GM_EB];     // tanh(h')
    float obsT[8][GM_EB];    // masked observations, [component][episode] (rows >= S stay zero)
    float logit[GM_EB][4];   // fc2 output per episode (A <= 4), written by lanes 0..15, read by every replica
    float bi[96], bh[96];    // gate biases (rows = gate * 32 + unit)
    float b1[32];
    float b2[4];
};

template <int S, int A>
struct GruMfma {
    static_assert(S % 4 == 0 && S <= 8, "fc1 runs in k-blocks of 4");
    static_assert(A <= 4, "fc2 outputs sit in rows 0..3 of one tile (lanes 0..15)");
    static constexpr int SC = S / 4;
    float w1[2][SC];         // A fragments: W1[16 t + li][4 c + lg]
    float wih[6][8];         // W_ih[16 m + li][4 kb + lg]
    float whh[6][8];
    float w2[8];             // W2[o = li][4 g + lg] (0 for li >= A)

    __device__ __forceinline__ void load(const float *__restrict__ theta, int lane, GruMfmaLds<S, A> &lds)
    {
        const int li = lane & 15, lg = lane >> 4;
        const float *p = theta;
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int c = 0; c < SC; ++c) w1[t][c] = p[(16 * t + li) * S + 4 * c + lg];
        p += H * S;
        if (lane < 32) lds.b1[lane] = p[lane];
        p += H;
        const float *pih = p, *phh = p + 3 * H * H, *pbi = p + 6 * H * H, *pbh = pbi + 3 * H;
#pragma unroll
        for (int m = 0; m < 6; ++m)
#pragma unroll
            for (int kb = 0; kb < 8; ++kb) {
                wih[m][kb] = pih[(16 * m + li) * H + 4 * kb + lg];
                whh[m][kb] = phh[(16 * m + li) * H + 4 * kb + lg];
            }
        for (int i = lane; i < 96; i += 64) {
            lds.bi[i] = pbi[i];
            lds.bh[i] = pbh[i];
        }
        p = pbh + 3 * H;
#pragma unroll
        for (int g = 0; g < 8; ++g) w2[g] = li < A ? p[li * H + 4 * g + lg] : 0.0f;
        if (lane < 4) lds.b2[lane] = lane < A ? p[A * H + lane] : 0.0f;
        // observation rows beyond S are never written: keep them zero
        for (int i = lane; i < 8 * GM_EB; i += 64) (&lds.obsT[0][0])[i] = 0.0f;
    }

    // the bias of rows [16 m + 4 lg, +4) as an accumulator fragment (same value in every column)
    __device__ static __forceinline__ gm_f32x4 bias_frag(const float *b, int m, int lg)
    {
        const float4 v = reinterpret_cast<const float4 *>(b)[4 * m + lg];
        return gm_f32x4{v.x, v.y, v.z, v.w};
    }

    // One time step for the 16 episode columns.  lds.obsT must hold the observations; hreg[t][r] is this lane's
    // copy of h[unit 16 t + 4 lg + r][episode li].  On return lds.logit[e] holds the fc2 outputs.
    __device__ __forceinline__ void step(const TanhEntry *tab, GruMfmaLds<S, A> &lds, float (&hreg)[2][4], int lane) const
    {
        const int li = lane & 15, lg = lane >> 4;
        // ---- fc1
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            gm_f32x4 d = bias_frag(lds.b1, t, lg);
#pragma unroll
            for (int c = 0; c < SC; ++c) d = mfma4(w1[t][c], (&lds.obsT[0][0])[64 * c + lane], d);
#pragma unroll
            for (int r = 0; r < 4; ++r) lds.aT[16 * t + 4 * lg + r][li] = tanh_(tab, d[r]);
        }
        wave_lds_sync();
        // ---- gate contractions: B fragments of a and h for the 8 k-blocks
        float xa[8], xh[8];
#pragma unroll
        for (int kb = 0; kb < 8; ++kb) {
            xa[kb] = (&lds.aT[0][0])[64 * kb + lane];
            xh[kb] = (&lds.hT[0][0])[64 * kb + lane];
        }
        wave_lds_sync();                                  // every lane holds its fragments: hT / yT may be rewritten
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            float ti[3][4], th[3][4];                     // gate (r, z, n) totals of unit 16 t + 4 lg + r
#pragma unroll
            for (int g = 0; g < 3; ++g) {
                const int m = 2 * g + t;                  // rows 32 g + 16 t ..
                gm_f32x4 il = bias_frag(lds.bi, m, lg), hl = bias_frag(lds.bh, m, lg);
                gm_f32x4 iu = {0.0f, 0.0f, 0.0f, 0.0f}, hu = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
                for (int kb = 0; kb < 4; ++kb) {          // four independent accumulation chains interleaved
                    il = mfma4(wih[m][kb], xa[kb], il);
                    hl = mfma4(whh[m][kb], xh[kb], hl);
                    iu = mfma4(wih[m][4 + kb], xa[4 + kb], iu);
                    hu = mfma4(whh[m][4 + kb], xh[4 + kb], hu);
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    ti[g][r] = il[r] + iu[r];
                    th[g][r] = hl[r] + hu[r];
                }
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float pr = 0.5f * (ti[0][r] + th[0][r]), pz = 0.5f * (ti[1][r] + th[1][r]);
                float ur, uz;
                const int32_t ir = tanh_index(pr, ur), iz = tanh_index(pz, uz);
                const TanhEntry er = tab[ir], ez = tab[iz];
                const float rg = fma_(0.5f, tanh_eval(er, ur, pr), 0.5f);
                const float zg = fma_(0.5f, tanh_eval(ez, uz, pz), 0.5f);
                const float ng = tanh_(tab, fma_(rg, th[2][r], ti[2][r]));
                const float hn = fma_(zg, hreg[t][r] - ng, ng);
                hreg[t][r] = hn;
                lds.hT[16 * t + 4 * lg + r][li] = hn;
                lds.yT[16 * t + 4 * lg + r][li] = tanh_(tab, hn);
            }
        }
        wave_lds_sync();
        // ---- fc2: one MFMA per group of 4 hidden units, tree over the 8 groups; rows 0..A-1 live in lanes 0..15
        gm_f32x4 pg[8];
#pragma unroll
        for (int g = 0; g < 8; ++g)
            pg[g] = mfma4(w2[g], (&lds.yT[0][0])[64 * g + lane], gm_f32x4{0.0f, 0.0f, 0.0f, 0.0f});
        if (lg == 0) {
            float out[4];
#pragma unroll
            for (int o = 0; o < 4; ++o)
                out[o] = (((pg[0][o] + pg[1][o]) + (pg[2][o] + pg[3][o])) + ((pg[4][o] + pg[5][o]) + (pg[6][o] + pg[7][o]))) +
                         lds.b2[o];
            *reinterpret_cast<float4 *>(&lds.logit[li][0]) = float4{out[0], out[1], out[2], out[3]};
        }
        wave_lds_sync();
    }
};

}  // namespace ses
