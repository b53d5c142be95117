// ses_policy.h -- GymEnvModel forward (networks/neural_network.py:20-36) for one env, with the
// 32 hidden units of the MLP spread over LPE adjacent lanes (LPE = 1, 2, 4 or 8).
//
// Why lanes-per-env: at the benchmark population (4096 offspring x 5 episodes = 20 480 envs) one
// lane per env would give 320 wavefronts for 1024 SIMDs.  Splitting the hidden layer over LPE lanes
// multiplies the wave count and divides the serial work per env-step (the 32 tanh evaluations are
// the bulk of it) at the cost of a DPP butterfly for the fc2 dot products and redundant physics.
//
// Canonical arithmetic (identical for every LPE, restated by oracle/ses_oracle.c):
//   fc1  : acc = b1[j]; acc = fma(W1[j][k], obs[k], acc) for k ascending; a[j] = tanh(acc)
//          (tanh = piecewise-cubic table in LDS, ses_math.h)
//   fc2  : 8 groups of 4 consecutive hidden units, in-order fma chain from the plain product;
//          balanced pairwise tree over the 8 group sums; + bias last
//   argmax: first maximum wins
// Per-offspring weights make the "population GEMM" a batch of 32xS matvecs with M = E = 5 columns
// per weight set; fp32 MFMA runs at the VALU rate on gfx950 and a 32x32x2 tile would be 84 %
// padding, so the contraction stays on the VALU with the weights resident in VGPRs.
#pragma once
#include <hip/hip_runtime.h>

#include "ses_math.h"

namespace ses {

constexpr int H = 32;

// DPP lane exchange; all lanes of the wave must be active
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
constexpr int DPP_QUAD_XOR1 = 0xB1;        // quad_perm [1,0,3,2]
constexpr int DPP_QUAD_XOR2 = 0x4E;        // quad_perm [2,3,0,1]
constexpr int DPP_ROW_HALF_MIRROR = 0x141; // lane i <-> 7-i inside each group of 8
constexpr int DPP_ROW_ROR8 = 0x128;        // lane i <- lane (i + 8) % 16 inside each row of 16
constexpr int DPP_ROW_ROR12 = 0x12C;       // row_ror:12: lane i <- lane (i + 4) % 16
constexpr int DPP_QUAD_00_22 = 0xA0;       // quad_perm [0,0,2,2]: odd lanes <- their even neighbour
constexpr int DPP_ROW_BCAST1 = 0x151;      // row_newbcast:1: every lane of a row <- lane 1 of the row
constexpr int DPP_ROW_NEWBCAST15 = 0x15F;    // row_newbcast:15: every lane of a row <- lane 15 of the row
constexpr int DPP_ROW_ROR4 = 0x124;        // row_ror:4: lane i <- lane (i + 12) % 16 = i - 4
constexpr int DPP_QUAD_BCAST0 = 0x00;      // quad_perm [0,0,0,0]: every lane of a quad <- its lane 0
constexpr int DPP_QUAD_BCAST1 = 0x55;      // quad_perm [1,1,1,1]
constexpr int DPP_QUAD_BCAST2 = 0xAA;      // quad_perm [2,2,2,2]
constexpr int DPP_QUAD_BCAST3 = 0xFF;      // quad_perm [3,3,3,3]

// sum over the LPE lanes that share one env; every lane ends with the same bits
template <int LPE>
__device__ __forceinline__ float lanes_sum(float v)
{
    if constexpr (LPE >= 2) v = v + dpp_mov<DPP_QUAD_XOR1>(v);
    if constexpr (LPE >= 4) v = v + dpp_mov<DPP_QUAD_XOR2>(v);
    if constexpr (LPE >= 8) v = v + dpp_mov<DPP_ROW_HALF_MIRROR>(v);  // both quads hold their sum already
    return v;
}

template <int S, int A, int LPE>
struct MlpSlice {
    static constexpr int U = H / LPE;   // hidden units owned by this lane
    static constexpr int G = U / 4;     // fc2 groups owned by this lane
    // LPE = 16 (U = 2): a lane pair owns one fc2 group; the group's in-order chain passes from the even lane to the odd
    // one through one DPP move, the tree over the 8 groups runs over the odd lanes of the row, and the action is
    // broadcast to the row (finish() below).  Same canonical arithmetic, 83 instead of 104 (LPE 8) VALU instructions
    // per step and wave: for populations that cannot fill the chip otherwise, and as the light wave of the mixed split.
    // LPE = 32 (U = 1, round 6): a lane QUAD owns one fc2 group.  Every lane of the quad keeps the group's four W2 columns and
    // fetches the three activations it does not own through the DPP operand of the multiply / fma itself (quad broadcasts:
    // no separate move), so all four lanes hold the group's in-order chain; the tree over the 8 groups -- one per quad, four
    // per 16-lane row -- runs inside the rows (row_ror 4, 8) and crosses the two rows of the env with one
    // v_permlane16_swap on two copies.  For populations of at most 2048 envs (<= 1024 waves at two envs per wave).
    static_assert(U % 4 == 0 || U == 2 || U == 1, "a lane owns whole fc2 groups, or a lane pair / quad owns one");
    static constexpr int W2N = U <= 2 ? 4 : U;      // W2 columns a lane keeps per output
    float w1[U][S];   // times 32 (the tanh table's 1/h), see tanh_index_scaled
    float b1[U];      // times 32
    float w2[A][W2N];
    float b2[A];

    // theta: this offspring's row; sub: lane index inside the env's lane group
    __device__ __forceinline__ void load(const float *__restrict__ theta, int sub)
    {
        const int j0 = sub * U;
        const float *pw1 = theta;
        const float *pb1 = theta + H * S;
        const float *pw2 = pb1 + H;
        const float *pb2 = pw2 + A * H;
#pragma unroll
        for (int u = 0; u < U; ++u) {
#pragma unroll
            for (int k = 0; k < S; ++k) w1[u][k] = SES_TANH_H_INV * pw1[(j0 + u) * S + k];
            b1[u] = SES_TANH_H_INV * pb1[j0 + u];
        }
#pragma unroll
        for (int a = 0; a < A; ++a) {
            if constexpr (U == 1) {
#pragma unroll
                for (int u = 0; u < 4; ++u) w2[a][u] = pw2[a * H + (j0 & ~3) + u];       // the whole group of this lane's quad
            } else if constexpr (U == 2) {
                // own columns first, then the columns of the pair's EVEN lane (what an odd lane multiplies its neighbour's
                // activations with; an even lane reads its own again, nobody uses its result)
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    w2[a][u] = pw2[a * H + j0 + u];
                    w2[a][2 + u] = pw2[a * H + (j0 & ~3) + u];
                }
            } else {
#pragma unroll
                for (int u = 0; u < U; ++u) w2[a][u] = pw2[a * H + j0 + u];
            }
            b2[a] = pb2[a];
        }
    }

    // The forward pass is split in two so that a fused kernel can put env work that does not depend on the
    // action between the table reads and their use:
    //   begin(): all pre-activations + table indices, then ALL ds_read_b128 back to back
    //   finish(): cubics, fc2, lane reduction.
    // (Left to itself hipcc keeps only two ds_read_b128 in flight and waits on them four times per step:
    //  22 % of the wave's cycles were SQ_WAIT_ANY.)
    struct Pending {
        float pre[U], frac[U];
        TanhEntry ent[U];
    };

    __device__ __forceinline__ void begin(const TanhEntry *tab, const float (&obs)[S], Pending &pd) const
    {
        int32_t idx[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            float acc = b1[u];
#pragma unroll
            for (int k = 0; k < S; ++k) acc = fma_(w1[u][k], obs[k], acc);
            pd.pre[u] = acc;                                   // 32 * pre-activation: only its sign is used below
            idx[u] = tanh_index_scaled(acc, pd.frac[u]);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < U; ++u) pd.ent[u] = tab[idx[u]];
    }

    __device__ __forceinline__ void finish(const Pending &pd, float (&logits)[A]) const
    {
        float a[U];
#pragma unroll
        for (int u = 0; u < U; ++u) a[u] = tanh_eval(pd.ent[u], pd.frac[u], pd.pre[u]);
        if constexpr (U == 1) {
#pragma unroll
            for (int o = 0; o < A; ++o) {
                // the group's canonical chain in every lane of the quad: a[4g + k] arrives through the DPP operand
                // (hipcc folds the move into the v_mul_f32_dpp / v_fmac_f32_dpp that uses it: no instruction of its own)
                float q = dpp_mov<DPP_QUAD_BCAST0>(a[0]) * w2[o][0];
                q = fma_(dpp_mov<DPP_QUAD_BCAST1>(a[0]), w2[o][1], q);
                q = fma_(dpp_mov<DPP_QUAD_BCAST2>(a[0]), w2[o][2], q);
                q = fma_(dpp_mov<DPP_QUAD_BCAST3>(a[0]), w2[o][3], q);
                // tree: quads of a row hold p0..p3 (row 0 of the env) / p4..p7 (row 1); commutativity of the IEEE add makes
                // (p[i+1] + p[i]) the bits of (p[i] + p[i+1])
                q = q + dpp_mov<DPP_ROW_ROR4>(q);                  // lanes 4-7: p1 + p0, lanes 12-15: p3 + p2
                q = q + dpp_mov<DPP_ROW_ROR8>(q);                  // lanes 12-15: (p3 + p2) + (p1 + p0)
                float lo = q, hi = q;
                asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 1" : "+v"(lo), "+v"(hi));
                q = lo + hi;                                       // lanes 12-15 of both rows: (p0..p3) + (p4..p7)
                logits[o] = dpp_mov<DPP_ROW_NEWBCAST15>(q + b2[o]);
            }
            return;
        }
        if constexpr (U == 2) {
#pragma unroll
            for (int o = 0; o < A; ++o) {
                // the ODD lane of a pair evaluates the group's whole in-order chain: units 4g, 4g + 1 live in its even
                // neighbour and arrive through the DPP operand of the multiply / fma (round 6: four instructions and four
                // dependent steps, where the even lane's partial chain + a move + the odd lane's half took five); w2[o][2..3]
                // are the even lane's columns.  (Even lanes compute something nobody reads.)
                float q = dpp_mov<DPP_QUAD_00_22>(a[0]) * w2[o][2];
                q = fma_(dpp_mov<DPP_QUAD_00_22>(a[1]), w2[o][3], q);
                q = fma_(w2[o][0], a[0], q);                       // odd lane: units 4g + 2, 4g + 3
                q = fma_(w2[o][1], a[1], q);
                // balanced tree over the 8 group sums, which live in the odd lanes 1, 3, ..., 15 of the row: lanes 1 and 3
                // (and 9, 11) end with the canonical ((p0+p1)+(p2+p3)) + ((p4+p5)+(p6+p7))
                q = q + dpp_mov<DPP_QUAD_XOR2>(q);
                q = q + dpp_mov<DPP_ROW_ROR12>(q);
                q = q + dpp_mov<DPP_ROW_ROR8>(q);
                logits[o] = dpp_mov<DPP_ROW_BCAST1>(q + b2[o]);    // lane 1 holds the logit: the whole row gets it
            }
            return;
        }
#pragma unroll
        for (int o = 0; o < A; ++o) {
            float p[G > 0 ? G : 1];
#pragma unroll
            for (int g = 0; g < G; ++g) {
                float acc = w2[o][4 * g] * a[4 * g];
                acc = fma_(w2[o][4 * g + 1], a[4 * g + 1], acc);
                acc = fma_(w2[o][4 * g + 2], a[4 * g + 2], acc);
                acc = fma_(w2[o][4 * g + 3], a[4 * g + 3], acc);
                p[g] = acc;
            }
            // in-lane levels of the balanced tree over the 8 groups
            float s;
            if constexpr (G == 8) {
                s = ((p[0] + p[1]) + (p[2] + p[3])) + ((p[4] + p[5]) + (p[6] + p[7]));
            } else if constexpr (G == 4) {
                s = (p[0] + p[1]) + (p[2] + p[3]);
            } else if constexpr (G == 2) {
                s = p[0] + p[1];
            } else {
                s = p[0];
            }
            logits[o] = lanes_sum<LPE>(s) + b2[o];
        }
    }

    // obs[S] -> logits[A] (identical in all LPE lanes of the env); tab: tanh table in LDS
    __device__ __forceinline__ void forward(const TanhEntry *tab, const float (&obs)[S], float (&logits)[A]) const
    {
        Pending pd;
        begin(tab, obs, pd);
        finish(pd, logits);
    }
};

// The same forward with the weights streamed from the offspring's (L2-resident) row instead of held in registers: for
// kernels whose env step dwarfs the policy (the Box2D envs) and that give an env only 1 or 2 lanes, where a lane's
// slice (32 / LPE units x (S + 1 + A) weights) would not fit the register file.  Same canonical arithmetic: the factor
// 32 is applied to W1 / b1 on the fly (exact), groups of 4 units chain in order, balanced tree over the 8 groups.
template <int S, int A, int LPE>
__device__ __forceinline__ void mlp_forward_streamed(const float *__restrict__ theta, int sub, const TanhEntry *tab,
                                                     const float (&obs)[S], float (&logits)[A])
{
    constexpr int U = H / LPE, G = U / 4;
    static_assert(U % 4 == 0, "a lane owns whole fc2 groups");
    const int j0 = sub * U;
    const float *pw1 = theta;
    const float *pb1 = theta + H * S;
    const float *pw2 = pb1 + H;
    const float *pb2 = pw2 + A * H;
    float p[A][G];
#pragma unroll
    for (int g = 0; g < G; ++g) {
        float a[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int j = j0 + 4 * g + u;
            float acc = SES_TANH_H_INV * pb1[j];
#pragma unroll
            for (int k = 0; k < S; ++k) acc = fma_(SES_TANH_H_INV * pw1[j * S + k], obs[k], acc);
            float frac;
            const int32_t idx = tanh_index_scaled(acc, frac);
            a[u] = tanh_eval(tab[idx], frac, acc);
        }
#pragma unroll
        for (int o = 0; o < A; ++o) {
            const float *w = pw2 + o * H + j0 + 4 * g;
            float acc = w[0] * a[0];
            acc = fma_(w[1], a[1], acc);
            acc = fma_(w[2], a[2], acc);
            acc = fma_(w[3], a[3], acc);
            p[o][g] = acc;
        }
        __builtin_amdgcn_sched_barrier(0);          // one group's loads in flight at a time: registers, not latency, are short
    }
#pragma unroll
    for (int o = 0; o < A; ++o) {
        float s;
        if constexpr (G == 8) {
            s = ((p[o][0] + p[o][1]) + (p[o][2] + p[o][3])) + ((p[o][4] + p[o][5]) + (p[o][6] + p[o][7]));
        } else if constexpr (G == 4) {
            s = (p[o][0] + p[o][1]) + (p[o][2] + p[o][3]);
        } else if constexpr (G == 2) {
            s = p[o][0] + p[o][1];
        } else {
            s = p[o][0];
        }
        logits[o] = lanes_sum<LPE>(s) + pb2[o];
    }
}

// cooperative copy of the tanh table into LDS; ends with a workgroup barrier
__device__ __forceinline__ void stage_tanh_table(TanhEntry *lds)
{
    const float4 *src = reinterpret_cast<const float4 *>(&SES_TANH_TABLE[0][0]);
    for (int i = threadIdx.x; i < SES_TANH_N; i += blockDim.x) {
        const float4 v = src[i];
        lds[i] = TanhEntry{v.x, v.y, v.z, v.w};
    }
    __syncthreads();
}

__device__ __forceinline__ int add_mask_bit(int v, unsigned long long mask)
{
    asm("v_addc_co_u32_e64 %0, vcc, %0, 0, %1" : "+v"(v) : "s"(mask) : "vcc");
    return v;
}

template <int A>
__device__ __forceinline__ int argmax_first(const float (&logits)[A])
{
    int best = 0;
    float bv = logits[0];
#pragma unroll
    for (int k = 1; k < A; ++k) {
        const bool gt = logits[k] > bv;
        best = gt ? k : best;
        bv = gt ? logits[k] : bv;
    }
    return best;
}

}  // namespace ses
