// ses_policy_pk.h -- the CartPole MLP step for a wave that has its SIMD to itself (round 6).
//
// Small per-GPU populations -- the 512 / 1024 offspring per GPU of the strong-scaling line, conf/cartpole.yaml's 97 -- are
// fewer waves than the chip has SIMDs: a rollout then costs max_step x the time ONE wave needs for a step.  A lone wave issues
// one instruction every ~2.2 ns whatever the instruction is (profiles/r01_valu_issue.txt: v_mul_f32 2.2 ns, v_pk_fma_f32 2.5 ns
// at one wave per SIMD; the two slots a packed instruction takes only show when several waves compete), and
// tools/chain_model.py shows the step of the 16-lanes-per-env loop to be bound by exactly that: 83 instructions issued in order
// take 207 ns (measured: 198) against a dependence chain of 144 ns.  So here -- and only here -- two IEEE operations per
// instruction pay: this file is the same canonical arithmetic (ses_policy.h, ses_cartpole.h) with
//   fc1      the lane's hidden units in pairs: (unit 2p, unit 2p + 1) advance with ONE v_pk_fma_f32 per input;
//   fc2      the two logits as a pair: one packed product / fma per hidden unit instead of two;
//   sin/cos  the two polynomials' Horner steps as a pair;
//   state    (x, theta) += tau (xd, thetad) and (xd, thetad) += tau (xacc, thetaacc) as pairs.
// Every half of a packed instruction is the IEEE operation the scalar form performs, in the same order: results are
// bit-identical (tests/test_gpu_parity.py and tools/fuzz_parity.py run both forms against the oracle).  The launcher
// (ses_rollout.hip::launch_cartpole_mlp) takes this form when the population gives every wave a SIMD of its own; with two
// or more waves per SIMD the scalar form is faster (round 1 measured packing at +1.5 ... +9 % there: NOTES.md).
#pragma once
#include "ses_cartpole.h"
#include "ses_policy.h"

namespace ses {

typedef float pk2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ pk2 pk_fma(pk2 a, pk2 b, pk2 c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ pk2 pk_splat(float v) { return pk2{v, v}; }

// MLP 4 -> 32 -> 2 with the hidden units spread over LPE = 8 or 16 adjacent lanes
template <int LPE>
struct MlpSlicePk {
    static constexpr int S = 4, A = 2;
    static constexpr int U = H / LPE;     // hidden units of this lane: 4 or 2
    static constexpr int UP = U / 2;      // ... in pairs
    static_assert(U == 2 || U == 4, "8 or 16 lanes per env");
    pk2 w1[UP][S];    // (W1[j0 + 2p][k], W1[j0 + 2p + 1][k]) x 32
    pk2 b1[UP];
    pk2 w2[U];        // (W2[0][j0 + u], W2[1][j0 + u]): the lane's own columns of both outputs
    pk2 w2e[2];       // U == 2: the columns of the pair's EVEN lane (see MlpSlice<.., 16>::finish)
    pk2 b2;

    __device__ __forceinline__ void load(const float *__restrict__ theta, int sub)
    {
        const int j0 = sub * U;
        const float *pw1 = theta, *pb1 = theta + H * S, *pw2 = pb1 + H, *pb2 = pw2 + A * H;
#pragma unroll
        for (int p = 0; p < UP; ++p) {
#pragma unroll
            for (int k = 0; k < S; ++k)
                w1[p][k] = pk2{SES_TANH_H_INV * pw1[(j0 + 2 * p) * S + k], SES_TANH_H_INV * pw1[(j0 + 2 * p + 1) * S + k]};
            b1[p] = pk2{SES_TANH_H_INV * pb1[j0 + 2 * p], SES_TANH_H_INV * pb1[j0 + 2 * p + 1]};
        }
#pragma unroll
        for (int u = 0; u < U; ++u) w2[u] = pk2{pw2[j0 + u], pw2[H + j0 + u]};
#pragma unroll
        for (int u = 0; u < 2; ++u) w2e[u] = pk2{pw2[(j0 & ~3) + u], pw2[H + (j0 & ~3) + u]};
        b2 = pk2{pb2[0], pb2[1]};
    }

    struct Pending {
        float pre[U], frac[U];
        TanhEntry ent[U];
    };

    __device__ __forceinline__ void begin(const TanhEntry *tab, const float (&obs)[S], Pending &pd) const
    {
        int32_t idx[U];
#pragma unroll
        for (int p = 0; p < UP; ++p) {
            pk2 acc = b1[p];
#pragma unroll
            for (int k = 0; k < S; ++k) acc = pk_fma(w1[p][k], pk_splat(obs[k]), acc);     // bias first, k ascending: canonical
            pd.pre[2 * p] = acc.x;
            pd.pre[2 * p + 1] = acc.y;
            idx[2 * p] = tanh_index_scaled(acc.x, pd.frac[2 * p]);
            idx[2 * p + 1] = tanh_index_scaled(acc.y, pd.frac[2 * p + 1]);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < U; ++u) pd.ent[u] = tab[idx[u]];
    }

    // (logit 0, logit 1), identical in all lanes of the env
    __device__ __forceinline__ pk2 finish(const Pending &pd) const
    {
        float a[U];
#pragma unroll
        for (int u = 0; u < U; ++u) a[u] = tanh_eval(pd.ent[u], pd.frac[u], pd.pre[u]);
        pk2 q;
        if constexpr (U == 2) {
            // odd lane of a pair: the group's in-order chain, units 4g, 4g + 1 from the even neighbour (two DPP moves: packed
            // instructions take no DPP operand), then its own 4g + 2, 4g + 3
            const float e0 = dpp_mov<DPP_QUAD_00_22>(a[0]), e1 = dpp_mov<DPP_QUAD_00_22>(a[1]);
            q = w2e[0] * pk_splat(e0);
            q = pk_fma(w2e[1], pk_splat(e1), q);
            q = pk_fma(w2[0], pk_splat(a[0]), q);
            q = pk_fma(w2[1], pk_splat(a[1]), q);
            float q0 = q.x, q1 = q.y;
            q0 = q0 + dpp_mov<DPP_QUAD_XOR2>(q0);
            q1 = q1 + dpp_mov<DPP_QUAD_XOR2>(q1);
            q0 = q0 + dpp_mov<DPP_ROW_ROR12>(q0);
            q1 = q1 + dpp_mov<DPP_ROW_ROR12>(q1);
            q0 = q0 + dpp_mov<DPP_ROW_ROR8>(q0);
            q1 = q1 + dpp_mov<DPP_ROW_ROR8>(q1);
            const pk2 lg = pk2{q0, q1} + b2;
            return pk2{dpp_mov<DPP_ROW_BCAST1>(lg.x), dpp_mov<DPP_ROW_BCAST1>(lg.y)};
        } else {
            // one fc2 group per lane: the chain over its four units, the tree over the 8 lanes of the env
            q = w2[0] * pk_splat(a[0]);
            q = pk_fma(w2[1], pk_splat(a[1]), q);
            q = pk_fma(w2[2], pk_splat(a[2]), q);
            q = pk_fma(w2[3], pk_splat(a[3]), q);
            const float q0 = lanes_sum<8>(q.x), q1 = lanes_sum<8>(q.y);
            return pk2{q0, q1} + b2;
        }
    }
};

// The step loop of one lone wave: fp32 dynamics, every lane's pole angle inside |th| <= SINCOS_SMALL_MAX (checked by the caller,
// wave-uniform).  Same control flow as rollout_cartpole_mlp_loop (ses_rollout.hip).
template <int LPE, bool FIXED_LENGTH, bool MASKED>
__device__ __forceinline__ void rollout_cartpole_mlp_loop_pk(const TanhEntry *tanh_tab, const MlpSlicePk<LPE> &net,
                                                             const float *s0, int max_step, uint32_t obs_mask, int &steps)
{
    pk2 P = {s0[0], s0[2]};                                   // (x, theta)
    pk2 V = {s0[1], s0[3]};                                   // (xd, thetad)
    const float th_clamp = register_constant(CP_TH_CLAMP), lim_clamp = register_constant(CP_CLAMP);
    const pk2 tau2 = pk_splat(register_constant(CP_TAU));
    // the two polynomials of sincos_small_ side by side: (cos, sin)
    const pk2 k0 = {register_constant(2.443315711809948e-5f), register_constant(-1.9515295891e-4f)};
    const pk2 k1 = {register_constant(-1.388731625493765e-3f), register_constant(8.3321608736e-3f)};
    const pk2 k2 = {register_constant(4.166664568298827e-2f), register_constant(-1.6666654611e-1f)};
    bool alive = true;
    unsigned long long alive_mask = ~0ull;
    for (int t = 0; t < max_step; ++t) {
        if constexpr (!FIXED_LENGTH) {
            if (__ballot(alive) == 0ull) break;
        }
        float obs[4] = {P.x, V.x, P.y, V.y};
        if constexpr (MASKED) {
#pragma unroll
            for (int k = 0; k < 4; ++k) obs[k] = ((obs_mask >> k) & 1u) ? 0.0f : obs[k];
        }
        typename MlpSlicePk<LPE>::Pending pending;
        net.begin(tanh_tab, obs, pending);
        // ---- the action-independent half of the physics (cartpole_pre_small), next to the table reads
        const float th = P.y, thd = V.y;
        const float z = th * th;
        const pk2 zz = pk_splat(z);
        pk2 pz = pk_fma(k0, zz, k1);
        pz = pk_fma(pz, zz, k2);
        pz = pz * zz;                                         // (pc * z, ps * z)
        CartPolePre pre;
        pre.sn = fma_(pz.y, th, th);
        pre.cs = fma_(pz.x, z, fma_(-0.5f, z, 1.0f));
        pre.q = CP_PML_OVER_MASS * (thd * thd);
        pre.gsn = CP_GRAVITY * pre.sn;
        pre.den = fma_(CP_DEN_C1, pre.cs * pre.cs, CP_DEN_C0);
        {
            const float r0 = __builtin_amdgcn_rcpf(pre.den);
            pre.rden = fma_(fma_(-pre.den, r0, 1.0f), r0, r0);
        }
        // ---- policy, action, the action-dependent half (cartpole_post)
        const pk2 lg = net.finish(pending);
        const bool one = lg.y > lg.x;                         // argmax_first<2>: the first maximum wins
        const float fom = one ? CP_FORCE_OVER_MASS : -CP_FORCE_OVER_MASS;
        const float temp = fma_(pre.q, pre.sn, fom);
        const float num = fma_(-pre.cs, temp, pre.gsn);
        const float thacc = cartpole_quotient(num, pre);
        const float xacc = fma_(-CP_PML_OVER_MASS * thacc, pre.cs, temp);
        const pk2 Pn = pk_fma(tau2, V, P);                    // (x + tau xd, theta + tau thetad): old velocities
        const pk2 Vn = pk_fma(tau2, pk2{xacc, thacc}, V);
        const float nx = clamp_sym_reg(Pn.x, lim_clamp);
        const float nth = __builtin_amdgcn_fmed3f(Pn.y, -th_clamp, th_clamp);
        const float nxd = clamp_sym_reg(Vn.x, lim_clamp);
        const float nthd = clamp_sym_reg(Vn.y, lim_clamp);
        if constexpr (FIXED_LENGTH) {
            P = pk2{nx, nth};
            V = pk2{nxd, nthd};
            steps = add_mask_bit(steps, alive_mask);
            alive_mask &= ~(__builtin_amdgcn_ballot_w64(__builtin_fabsf(nx) > CP_X_LIMIT) |
                            __builtin_amdgcn_ballot_w64(__builtin_fabsf(nth) > CP_THETA_LIMIT));
        } else {
            const bool term = (nx < -CP_X_LIMIT) || (nx > CP_X_LIMIT) || (nth < -CP_THETA_LIMIT) || (nth > CP_THETA_LIMIT);
            P = pk2{alive ? nx : P.x, alive ? nth : P.y};     // a finished env is frozen
            V = pk2{alive ? nxd : V.x, alive ? nthd : V.y};
            steps += (int)alive;
            alive = alive & !term;
        }
    }
}

}  // namespace ses
