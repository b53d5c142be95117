// ses_rollout.hip -- the rollout side of the hot path:
//   k_rollout_cartpole_mlp : RolloutWorker (loop.py:108-125) for a whole shard, one kernel
//   k_fitness_mean         : total_reward / eval_ep_num (loop.py:124)
//   k_env_step_cartpole    : standalone SoA env.step (gym_wrapper.py:32-45), the HBM-roofline kernel
//   k_policy_forward_mlp   : standalone population-batched GymEnvModel.forward (neural_network.py:20-36)
#include "ses_cartpole.h"
#include "ses_gru.h"
#include "ses_gru_lockstep.h"
#include "ses_gru_mfma.h"
#include "ses_gru_mfma4.h"
#include "ses_lander.h"
#include "ses_walker.h"
#include "ses_internal.h"
#include "ses_policy.h"
#include "ses_policy_pk.h"
#include "ses_spread.h"

namespace ses {

// ------------------------------------------------------------------------------------------------
// Fused rollout.  Thread layout: LPE adjacent lanes share one env; envs are numbered
// env = row * E + episode, so the E episodes of an offspring sit next to each other and their weight
// loads hit the same cache lines.  Each wave is its own 64-thread workgroup: 20 480 envs x LPE=4 gives
// 1280 independent workgroups that the dispatcher spreads over the 8 XCDs / 256 CUs; nothing is shared
// between workgroups, so no XCD-aware remap is needed.
// All state (4 floats of physics, the lane's slice of the weights, step counter) stays in VGPRs for
// the whole episode; HBM is touched once at the start (theta row, initial state) and once at the end.
// fp32 (default, folded constants) or gym-order float64 dynamics behind one interface
template <bool PHYS64>
struct CartPoleSim;

template <>
struct CartPoleSim<false> {
    CartPoleState st;
    CartPolePre pre;
    float th_clamp, lim_clamp;
    __device__ __forceinline__ void init(const float *s0)
    {
        st = CartPoleState{s0[0], s0[1], s0[2], s0[3]};
        th_clamp = register_constant(CP_TH_CLAMP);
        lim_clamp = register_constant(CP_CLAMP);
    }
    __device__ __forceinline__ void observe(float (&o)[4]) const { o[0] = st.x; o[1] = st.xd; o[2] = st.th; o[3] = st.thd; }
    // |pole angle| <= SINCOS_SMALL_MAX for this lane now -- and then for the whole episode (CP_TH_CLAMP)
    __device__ __forceinline__ bool small_angle() const { return __builtin_fabsf(st.th) <= SINCOS_SMALL_MAX; }
    template <bool SMALL>
    __device__ __forceinline__ void prepare()                               // action-independent half
    {
        pre = SMALL ? cartpole_pre_small(st) : cartpole_pre(st);
    }
    // wave mask of the lanes whose CURRENT state is terminal (== what advance() just returned when nothing was
    // frozen); two ballots of plain compares so that the mask is formed in SGPRs without a detour through a VGPR
    __device__ __forceinline__ unsigned long long terminal_mask() const
    {
        return __builtin_amdgcn_ballot_w64(__builtin_fabsf(st.x) > CP_X_LIMIT) |
               __builtin_amdgcn_ballot_w64(__builtin_fabsf(st.th) > CP_THETA_LIMIT);
    }
    __device__ __forceinline__ bool advance(int action, bool keep_old)
    {
        CartPoleState ns = st;
        const bool term = cartpole_post(ns, pre, action, th_clamp, lim_clamp);
        st.x = keep_old ? st.x : ns.x;
        st.xd = keep_old ? st.xd : ns.xd;
        st.th = keep_old ? st.th : ns.th;
        st.thd = keep_old ? st.thd : ns.thd;
        return term;
    }
};

template <>
struct CartPoleSim<true> {
    CartPoleState64 st;
    __device__ __forceinline__ void init(const float *s0) { st = CartPoleState64{s0[0], s0[1], s0[2], s0[3]}; }
    __device__ __forceinline__ void observe(float (&o)[4]) const
    {
        o[0] = (float)st.x; o[1] = (float)st.xd; o[2] = (float)st.th; o[3] = (float)st.thd;   // neural_network.py:22
    }
    __device__ __forceinline__ bool small_angle() const { return false; }
    template <bool SMALL>
    __device__ __forceinline__ void prepare() {}
    __device__ __forceinline__ unsigned long long terminal_mask() const
    {
        const double thr = 12 * 2 * 3.141592653589793 / 360;
        return __builtin_amdgcn_ballot_w64(__builtin_fabs(st.x) > 2.4) |
               __builtin_amdgcn_ballot_w64(__builtin_fabs(st.th) > thr);
    }
    __device__ __forceinline__ bool advance(int action, bool keep_old)
    {
        CartPoleState64 ns = st;
        const bool term = cartpole_step64(ns, action);
        st.x = keep_old ? st.x : ns.x;
        st.xd = keep_old ? st.xd : ns.xd;
        st.th = keep_old ? st.th : ns.th;
        st.thd = keep_old ? st.thd : ns.thd;
        return term;
    }
};

// The step loop of one wave.  MASKED: some observation components are zeroed (POMDP).  SMALL: every lane's pole
// angle is inside |th| <= SINCOS_SMALL_MAX, so the sin/cos argument reduction is skipped (ses_cartpole.h).
template <int LPE, bool FIXED_LENGTH, bool PHYS64, bool MASKED, bool SMALL>
__device__ __forceinline__ void rollout_cartpole_mlp_loop(const TanhEntry *tanh_tab, const MlpSlice<4, 2, LPE> &net,
                                                          CartPoleSim<PHYS64> &sim, int max_step, uint32_t obs_mask,
                                                          int &steps)
{
    bool alive = true;
    unsigned long long alive_mask = ~0ull;
    for (int t = 0; t < max_step; ++t) {
        if constexpr (!FIXED_LENGTH) {
            if (__ballot(alive) == 0ull) break;  // wave-uniform: every env of this wave is done
        }
        float obs[4];
        sim.observe(obs);
        if constexpr (MASKED) {
#pragma unroll
            for (int k = 0; k < 4; ++k) obs[k] = ((obs_mask >> k) & 1u) ? 0.0f : obs[k];
        }
        float logits[2];
        typename MlpSlice<4, 2, LPE>::Pending pending;
        net.begin(tanh_tab, obs, pending);                 // fc1 + tanh table reads in flight ...
        sim.template prepare<SMALL>();                     // ... next to the action-independent half of the physics
        net.finish(pending, logits);
        const int action = argmax_first<2>(logits);
        const bool term = sim.advance(action, FIXED_LENGTH ? false : !alive);  // episodic: a finished env is frozen
        // (the horizon needs no test here: a live env reaches max_step exactly when the loop ends)
        if constexpr (FIXED_LENGTH) {                      // the flag lives in SGPRs as a wave mask
            steps = add_mask_bit(steps, alive_mask);
            alive_mask &= ~sim.terminal_mask();
        } else {                                           // per-lane flag: it also freezes the env
            steps += (int)alive;
            alive = alive & !term;
        }
    }
}

// One wave's share of the rollout: envs [env0 + wave_local_index ...), LPE lanes per env.
// PK: the packed form of the step for a wave that has its SIMD to itself (ses_policy_pk.h; LPE 8 or 16, fp32 dynamics).
template <int LPE, bool FIXED_LENGTH, bool PHYS64 = false, bool PK = false>
__device__ __forceinline__ void rollout_cartpole_mlp_body(const TanhEntry *tanh_tab, long long lane_index, int env0,
                                                          const float *__restrict__ theta,
                                                          const float *__restrict__ init, int init_per_offspring,
                                                          int n_env, int E, int P, int max_step, uint32_t obs_mask,
                                                          double *__restrict__ ep_return,
                                                          int32_t *__restrict__ ep_steps)
{
    static_assert(!PK || (!PHYS64 && (LPE == 8 || LPE == 16)), "the packed step exists for 8 / 16 lanes per env, fp32 dynamics");
    int env = env0 + (int)(lane_index / LPE);
    const int sub = (int)(lane_index % LPE);
    const bool valid = env < n_env;
    env = valid ? env : n_env - 1;  // keep every lane active (DPP needs full waves); only valid lanes store
    const int row = env / E;
    const int ep = env - row * E;
    const float *s0 = init + ((size_t)(init_per_offspring ? row : 0) * E + ep) * 4;
    int steps = 0;
    // Loop variants, chosen once per wave (both conditions are wave-uniform):
    //  * fully observed envs (obs_mask == 0, a kernel argument) skip the four masking selects per step;
    //  * when every lane starts inside |th| <= 0.78 -- resets are U(-0.05, 0.05) -- the angle stays there for the
    //    whole episode (CP_TH_CLAMP) and the sin/cos argument reduction is skipped, bit-identically (sincos_small_);
    //    a caller-supplied initial state outside that range runs the general loop.
    const bool small = !PHYS64 && __ballot(!(__builtin_fabsf(s0[2]) <= SINCOS_SMALL_MAX)) == 0ull;
    bool done = false;
    if constexpr (PK) {
        if (small) {
            MlpSlicePk<LPE> netp;
            netp.load(theta + (size_t)row * P, sub);
            if (obs_mask == 0u)
                rollout_cartpole_mlp_loop_pk<LPE, FIXED_LENGTH, false>(tanh_tab, netp, s0, max_step, obs_mask, steps);
            else
                rollout_cartpole_mlp_loop_pk<LPE, FIXED_LENGTH, true>(tanh_tab, netp, s0, max_step, obs_mask, steps);
            done = true;
        }
    }
    if (!done) {
        MlpSlice<4, 2, LPE> net;
        net.load(theta + (size_t)row * P, sub);
        CartPoleSim<PHYS64> sim;
        sim.init(s0);
        if (obs_mask == 0u && small)
            rollout_cartpole_mlp_loop<LPE, FIXED_LENGTH, PHYS64, false, true>(tanh_tab, net, sim, max_step, obs_mask, steps);
        else if (small)
            rollout_cartpole_mlp_loop<LPE, FIXED_LENGTH, PHYS64, true, true>(tanh_tab, net, sim, max_step, obs_mask, steps);
        else
            rollout_cartpole_mlp_loop<LPE, FIXED_LENGTH, PHYS64, true, false>(tanh_tab, net, sim, max_step, obs_mask, steps);
    }
    if (valid && sub == 0) {
        if (ep_return) ep_return[env] = (double)steps;  // CartPole reward is 1 per step incl. the terminal one
        if (ep_steps) ep_steps[env] = steps;
    }
}

template <int LPE, bool FIXED_LENGTH, int BLOCK, bool PHYS64 = false, bool PK = false>
__global__ __launch_bounds__(BLOCK) void k_rollout_cartpole_mlp(const float *__restrict__ theta,
                                                             const float *__restrict__ init, int init_per_offspring,
                                                             int n_rows, int E, int P, int max_step,
                                                             uint32_t obs_mask, double *__restrict__ ep_return,
                                                             int32_t *__restrict__ ep_steps)
{
    __shared__ TanhEntry tanh_tab[SES_TANH_N];
    stage_tanh_table(tanh_tab);
    rollout_cartpole_mlp_body<LPE, FIXED_LENGTH, PHYS64, PK>(tanh_tab, (long long)blockIdx.x * BLOCK + threadIdx.x, 0, theta, init,
                                                 init_per_offspring, n_rows * E, E, P, max_step, obs_mask, ep_return,
                                                 ep_steps);
}

// Mixed split for mid-sized populations.  A lone wave issues one VALU instruction per 4 cycles, so a SIMD
// holding a single wave runs at half its issue rate, and with 20 480 envs neither split fills the 1024 SIMDs
// evenly (LPE 4: 1280 waves, a quarter of the SIMDs carry two; LPE 8: 2560 waves, half carry three).
// Here the first `waves8` single-wave workgroups take 8 envs each at 8 lanes per env -- the dispatcher deals
// them one per SIMD -- and the remaining envs follow at 4 lanes per env, so every SIMD ends up with one light
// wave (151 instructions per step) and at most one heavy wave (218).  Results do not depend on the split:
// every LPE variant evaluates the same canonical arithmetic.
// Round 3: the light wave may also run at 16 lanes per env (LIGHT = 16: 4 envs, 83 instructions per step).  At the
// benchmark population -- 20 480 envs -- 1024 such waves + 1024 waves at 4 lanes per env put exactly one light and one
// heavy wave, 20 envs and 243 instructions per step on EVERY SIMD (LIGHT = 8: 264 on three quarters of them, a lone
// light wave on the rest).  launch_cartpole_mlp picks the split whose busiest SIMD has the least to issue.
// Round 6: the second kind of wave is a template argument too.  REST = 4: as above.  (FIRST, REST) = (8, 16): populations between
// one and one and a half waves per SIMD at 8 lanes per env (8192 < envs <= 12 288: the 2048 offspring per GPU of the strong line at
// two GPUs) -- 1024 waves of 8 envs, one per SIMD, then the remaining envs four to a wave: a doubled SIMD issues 104 + 83 instructions
// per step where the pure split gave a quarter of the SIMDs 2 x 104.
template <bool FIXED_LENGTH, int LIGHT, int REST = 4>
__global__ __launch_bounds__(64) void k_rollout_cartpole_mlp_mix(const float *__restrict__ theta,
                                                                 const float *__restrict__ init,
                                                                 int init_per_offspring, int n_rows, int E, int P,
                                                                 int max_step, uint32_t obs_mask, int waves_light,
                                                                 double *__restrict__ ep_return,
                                                                 int32_t *__restrict__ ep_steps)
{
    constexpr int EPW = 64 / LIGHT;                    // envs of a wave of the first kind
    __shared__ TanhEntry tanh_tab[SES_TANH_N];
    stage_tanh_table(tanh_tab);
    const int n_env = n_rows * E;
    if ((int)blockIdx.x < waves_light) {
        rollout_cartpole_mlp_body<LIGHT, FIXED_LENGTH>(tanh_tab, (long long)blockIdx.x * 64 + threadIdx.x, 0, theta, init,
                                                       init_per_offspring, n_env < waves_light * EPW ? n_env : waves_light * EPW,
                                                       E, P, max_step, obs_mask, ep_return, ep_steps);
    } else {
        rollout_cartpole_mlp_body<REST, FIXED_LENGTH>(tanh_tab, (long long)(blockIdx.x - waves_light) * 64 + threadIdx.x,
                                                      waves_light * EPW, theta, init, init_per_offspring, n_env, E, P, max_step,
                                                      obs_mask, ep_return, ep_steps);
    }
}

// GRU policy: one offspring per wavefront (ses_gru.h), its E episodes run one after the other so the
// 6 x 16 + ... weights per lane stay in VGPRs across all of them.  4 offspring per 256-thread workgroup
// share one copy of the tanh table; the waves never synchronise with each other (episode lengths differ).
template <bool FIXED_LENGTH>
__global__ __launch_bounds__(256) void k_rollout_cartpole_gru(const float *__restrict__ theta,
                                                              const float *__restrict__ init, int init_per_offspring,
                                                              int n_rows, int E, int P, int max_step,
                                                              uint32_t obs_mask, double *__restrict__ ep_return,
                                                              int32_t *__restrict__ ep_steps, int ep_parallel)
{
    __shared__ TanhEntry tanh_tab[SES_TANH_N];
    __shared__ __attribute__((aligned(16))) float vecs[4][64];
    stage_tanh_table(tanh_tab);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    // ep_parallel: one wave per (offspring, episode) instead of one per offspring -- for populations too small to
    // fill the chip the E episodes of an offspring run side by side and the rollout takes one episode's time
    const int unit = blockIdx.x * 4 + wave, n_units = ep_parallel ? n_rows * E : n_rows;
    const bool valid = unit < n_units;
    const int u = valid ? unit : n_units - 1;
    const int row = ep_parallel ? u / E : u;
    const int ep_begin = ep_parallel ? u - row * E : 0, ep_end = ep_parallel ? ep_begin + 1 : E;
    GruSlice<4, 2> net;
    net.load(theta + (size_t)row * P, lane);
    float *vec = vecs[wave];
    for (int ep = ep_begin; ep < ep_end; ++ep) {
        const float *s0 = init + ((size_t)(init_per_offspring ? row : 0) * E + ep) * 4;
        CartPoleState st{s0[0], s0[1], s0[2], s0[3]};
        float h = 0.0f;                                   // GymEnvModel.reset(), neural_network.py:38-40
        wave_lds_sync();
        if (lane < 32) vec[2 * lane + 1] = 0.0f;
        wave_lds_sync();
        int steps = 0;
        bool alive = true;
        for (int t = 0; t < max_step; ++t) {
            if constexpr (!FIXED_LENGTH) {
                if (__builtin_amdgcn_readfirstlane((int)alive) == 0) break;   // one env per wave: uniform
            }
            float obs[4];
            obs[0] = (obs_mask & 1u) ? 0.0f : st.x;
            obs[1] = (obs_mask & 2u) ? 0.0f : st.xd;
            obs[2] = (obs_mask & 4u) ? 0.0f : st.th;
            obs[3] = (obs_mask & 8u) ? 0.0f : st.thd;
            float logits[2];
            net.forward(tanh_tab, obs, h, vec, lane, logits);
            const int action = argmax_first<2>(logits);
            CartPoleState ns = st;
            const bool term = cartpole_step_general(ns, action);
            const bool advance = FIXED_LENGTH ? true : alive;
            st.x = advance ? ns.x : st.x;
            st.xd = advance ? ns.xd : st.xd;
            st.th = advance ? ns.th : st.th;
            st.thd = advance ? ns.thd : st.thd;
            const int nsteps = steps + 1;
            const bool finished = term | (nsteps >= max_step);
            steps = alive ? nsteps : steps;
            alive = alive & !finished;
        }
        if (valid && lane == 0) {
            if (ep_return) ep_return[(size_t)row * E + ep] = (double)steps;
            if (ep_steps) ep_steps[(size_t)row * E + ep] = steps;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Lockstep GRU rollout (ses_gru_lockstep.h): one offspring per wave, up to 8 episodes advance together,
// lane l owns the env of episode (l & 7).  EnvT adapts an env to the kernel.
struct CartPoleLs {
    static constexpr int S = 4, A = 2, INIT_W = 4;
    struct State {
        CartPoleState st;
    };
    __device__ static __forceinline__ void reset(State &s, const float *__restrict__ u, int)
    {
        s.st = CartPoleState{u[0], u[1], u[2], u[3]};
    }
    __device__ static __forceinline__ void observe(const State &s, float (&obs)[S])
    {
        obs[0] = s.st.x; obs[1] = s.st.xd; obs[2] = s.st.th; obs[3] = s.st.thd;
    }
    // advance with the policy output; returns the reward, sets done.  `freeze`: keep the old state (finished env)
    __device__ static __forceinline__ float step(State &s, const float (&logits)[A], const TanhEntry *, bool freeze,
                                                 bool &done)
    {
        const int action = argmax_first<A>(logits);
        CartPoleState ns = s.st;
        done = cartpole_step_general(ns, action);
        s.st.x = freeze ? s.st.x : ns.x;
        s.st.xd = freeze ? s.st.xd : ns.xd;
        s.st.th = freeze ? s.st.th : ns.th;
        s.st.thd = freeze ? s.st.thd : ns.thd;
        return 1.0f;
    }
};

struct CartPoleLs64 {
    static constexpr int S = 4, A = 2, INIT_W = 4;
    struct State {
        CartPoleState64 st;
    };
    __device__ static __forceinline__ void reset(State &s, const float *__restrict__ u, int)
    {
        s.st = CartPoleState64{u[0], u[1], u[2], u[3]};
    }
    __device__ static __forceinline__ void observe(const State &s, float (&obs)[S])
    {
        obs[0] = (float)s.st.x; obs[1] = (float)s.st.xd; obs[2] = (float)s.st.th; obs[3] = (float)s.st.thd;
    }
    __device__ static __forceinline__ float step(State &s, const float (&logits)[A], const TanhEntry *, bool freeze,
                                                 bool &done)
    {
        const int action = argmax_first<A>(logits);
        CartPoleState64 ns = s.st;
        done = cartpole_step64(ns, action);
        s.st.x = freeze ? s.st.x : ns.x;
        s.st.xd = freeze ? s.st.xd : ns.xd;
        s.st.th = freeze ? s.st.th : ns.th;
        s.st.thd = freeze ? s.st.thd : ns.thd;
        return 1.0f;
    }
};

struct LanderLs {
    static constexpr int S = 8, A = 4, INIT_W = 16;
    struct State {
        LanderState st;
    };
    __device__ static __forceinline__ void reset(State &s, const float *__restrict__ u, int slot)
    {
        __shared__ float terrain[4][32][LL_TERRAIN_ROW];       // one terrain row per (wave, env slot); slot < 32
        ll_reset(s.st, u, terrain[threadIdx.x >> 6][slot]);
    }
    __device__ static __forceinline__ void observe(const State &s, float (&obs)[S]) { ll_obs(s.st, obs); }
    __device__ static __forceinline__ float step(State &s, const float (&logits)[A], const TanhEntry *tab, bool freeze,
                                                 bool &done)
    {
        const float a0 = tanh_(tab, logits[0]), a1 = tanh_(tab, logits[1]);
        float r = 0.0f;
        done = true;
        if (!freeze) r = ll_step(s.st, a0, a1, done);          // a finished env is frozen (the env code has no wave votes)
        return r;
    }
};

template <typename EnvT, bool FIXED_LENGTH, int NP, bool ODD>
__device__ __forceinline__ void gru_lockstep_batch(const TanhEntry *tanh_tab, GruLockstepLds<EnvT::S, EnvT::A> &lds,
                                                   const GruLockstep<EnvT::S, EnvT::A> &net, int lane, int nb,
                                                   const float *__restrict__ init_rows, int max_step, uint32_t obs_mask,
                                                   double *__restrict__ ret_out, int32_t *__restrict__ steps_out,
                                                   bool valid_row)
{
    constexpr int S = EnvT::S, A = EnvT::A;
    const int slot = lane & 7;
    const bool owner_valid = slot < nb;
    typename EnvT::State st;
    EnvT::reset(st, init_rows + (size_t)(owner_valid ? slot : 0) * EnvT::INIT_W, slot);   // padding slots replay episode 0
    float hreg[NP];
#pragma unroll
    for (int p = 0; p < NP; ++p) hreg[p] = 0.0f;                                  // GymEnvModel.reset()
    wave_lds_sync();
    if (lane < 32) {
#pragma unroll
        for (int e = 0; e < GL_EB; ++e) lds.ah[e][lane][1] = 0.0f;
    }
    double ret = 0.0;
    int steps = 0;
    bool alive = true;
    for (int t = 0; t < max_step; ++t) {
        if constexpr (!FIXED_LENGTH) {
            if (__ballot(alive & owner_valid) == 0ull) break;
        }
        float obs[S];
        EnvT::observe(st, obs);
        if (lane < GL_EB) {
#pragma unroll
            for (int k = 0; k < S; ++k) lds.obs[lane][k] = ((obs_mask >> k) & 1u) ? 0.0f : obs[k];
        }
        wave_lds_sync();
        net.template step<NP, ODD>(tanh_tab, lds, hreg, lane);
        float logits[A];
        net.logits_of(lds, lane, logits);
        bool term;
        const bool freeze = FIXED_LENGTH ? false : !alive;
        const float r = EnvT::step(st, logits, tanh_tab, freeze, term);
        const int nsteps = steps + 1;
        const bool finished = term | (nsteps >= max_step);
        ret = alive ? ret + (double)r : ret;
        steps = alive ? nsteps : steps;
        alive = alive & !finished;
    }
    if (valid_row && lane < GL_EB && owner_valid) {
        if (ret_out) ret_out[slot] = ret;
        if (steps_out) steps_out[slot] = steps;
    }
}

// WAVES offspring per workgroup (they share one copy of the tanh table and never synchronise).  CartPole: 4.
// LunarLander: 1 -- a wave lives as long as the longest of its episodes, a workgroup as long as its longest wave, and
// with a 20 000-instruction env step the tail is what the kernel time is made of: single-wave workgroups free their
// SIMD slot for the next offspring as soon as their own five episodes are over.
template <typename EnvT, bool FIXED_LENGTH, int WAVES>
__global__ __launch_bounds__(64 * WAVES, 2) void k_rollout_gru_lockstep(const float *__restrict__ theta,
                                                              const float *__restrict__ init, int init_per_offspring,
                                                              int n_rows, int E, int P, int max_step, uint32_t obs_mask,
                                                              double *__restrict__ ep_return,
                                                              int32_t *__restrict__ ep_steps)
{
    __shared__ TanhEntry tanh_tab[SES_TANH_N];
    __shared__ __attribute__((aligned(16))) GruLockstepLds<EnvT::S, EnvT::A> ldsv[WAVES];
    stage_tanh_table(tanh_tab);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    int row = blockIdx.x * WAVES + wave;
    const bool valid = row < n_rows;
    row = valid ? row : n_rows - 1;
    GruLockstepLds<EnvT::S, EnvT::A> &lds = ldsv[wave];
    GruLockstep<EnvT::S, EnvT::A> net;
    net.load(theta + (size_t)row * P, lane, lds);
    wave_lds_sync();
    for (int e0 = 0; e0 < E; e0 += GL_EB) {
        const int nb = E - e0 < GL_EB ? E - e0 : GL_EB;
        const float *rows = init + ((size_t)(init_per_offspring ? row : 0) * E + e0) * EnvT::INIT_W;
        double *ro = ep_return ? ep_return + (size_t)row * E + e0 : nullptr;
        int32_t *so = ep_steps ? ep_steps + (size_t)row * E + e0 : nullptr;
#define SES_LS_CASE(NP_, ODD_)                                                                                        \
    gru_lockstep_batch<EnvT, FIXED_LENGTH, NP_, ODD_>(tanh_tab, lds, net, lane, nb, rows, max_step, obs_mask, ro, so, valid)
        switch (nb) {
            case 1: SES_LS_CASE(1, true); break;
            case 2: SES_LS_CASE(1, false); break;
            case 3: SES_LS_CASE(2, true); break;
            case 4: SES_LS_CASE(2, false); break;
            case 5: SES_LS_CASE(3, true); break;
            case 6: SES_LS_CASE(3, false); break;
            case 7: SES_LS_CASE(4, true); break;
            default: SES_LS_CASE(4, false); break;
        }
#undef SES_LS_CASE
    }
}

// ------------------------------------------------------------------------------------------------
// The lockstep GRU rollout with the policy step on v_mfma_f32_4x4x1_16b_f32 (ses_gru_mfma4.h, round 6): one offspring per
// wave, up to 8 episodes, lane l owns the env of episode (l & 7) exactly as in gru_lockstep_batch; only the policy differs.
template <typename EnvT, bool FIXED_LENGTH>
__device__ __forceinline__ void gru_mfma4_batch(const TanhEntry *tanh_tab, GruMfma4Lds<EnvT::S, EnvT::A> &lds,
                                                const GruMfma4<EnvT::S, EnvT::A> &net, int lane, int nb,
                                                const float *__restrict__ init_rows, int max_step, uint32_t obs_mask,
                                                double *__restrict__ ret_out, int32_t *__restrict__ steps_out, bool valid_row)
{
    constexpr int S = EnvT::S, A = EnvT::A;
    const int slot = lane & 7;
    const bool owner_valid = slot < nb;
    typename EnvT::State st;
    EnvT::reset(st, init_rows + (size_t)(owner_valid ? slot : 0) * EnvT::INIT_W, slot);   // padding slots replay episode 0
    float hreg[4] = {0.0f, 0.0f, 0.0f, 0.0f};                                     // GymEnvModel.reset()
    wave_lds_sync();
    if (lane < 32) {
#pragma unroll
        for (int e = 0; e < G4_EB; ++e) lds.ah[e][lane][1] = 0.0f;
    }
    double ret = 0.0;
    int steps = 0;
    bool alive = true;
    for (int t = 0; t < max_step; ++t) {
        if constexpr (!FIXED_LENGTH) {
            if (__ballot(alive & owner_valid) == 0ull) break;
        }
        float obs[S];
        EnvT::observe(st, obs);
        if (lane < G4_EB) {
#pragma unroll
            for (int k = 0; k < S; ++k) lds.obs[lane][k] = ((obs_mask >> k) & 1u) ? 0.0f : obs[k];
        }
        wave_lds_sync();
        net.step(tanh_tab, lds, hreg, lane);
        float logits[A];
        GruMfma4<S, A>::logits_of(lds, lane, logits);
        bool term;
        const bool freeze = FIXED_LENGTH ? false : !alive;
        const float r = EnvT::step(st, logits, tanh_tab, freeze, term);
        const int nsteps = steps + 1;
        const bool finished = term | (nsteps >= max_step);
        ret = alive ? ret + (double)r : ret;
        steps = alive ? nsteps : steps;
        alive = alive & !finished;
    }
    if (valid_row && lane < G4_EB && owner_valid) {
        if (ret_out) ret_out[slot] = ret;
        if (steps_out) steps_out[slot] = steps;
    }
}

// 4 offspring per 256-thread workgroup (one copy of the tanh table), two waves per SIMD: W_hh (96 values per lane) in registers,
// W_ih in the wave's LDS block (13.5 KB; two workgroups = 8 waves fill 158 of the CU's 160 KB).  With all 192 gate weights in
// registers the kernel needed 360 of them, i.e. one wave per SIMD, and a lone wave ran the step's ~550 non-MFMA instructions
// with every stall exposed: 4.01 ms against this form's -- see DESIGN.md section 4.
template <typename EnvT, bool FIXED_LENGTH>
__global__ __launch_bounds__(256, 2) void k_rollout_gru_mfma4(const float *__restrict__ theta, const float *__restrict__ init,
                                                              int init_per_offspring, int n_rows, int E, int P, int max_step,
                                                              uint32_t obs_mask, double *__restrict__ ep_return,
                                                              int32_t *__restrict__ ep_steps)
{
    __shared__ TanhEntry tanh_tab[SES_TANH_N];
    __shared__ __attribute__((aligned(16))) GruMfma4Lds<EnvT::S, EnvT::A> ldsv[4];
    stage_tanh_table(tanh_tab);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    int row = blockIdx.x * 4 + wave;
    const bool valid = row < n_rows;
    row = valid ? row : n_rows - 1;
    GruMfma4Lds<EnvT::S, EnvT::A> &lds = ldsv[wave];
    GruMfma4<EnvT::S, EnvT::A> net;
    net.load(theta + (size_t)row * P, lane, lds);
    wave_lds_sync();
    for (int e0 = 0; e0 < E; e0 += G4_EB) {
        const int nb = E - e0 < G4_EB ? E - e0 : G4_EB;
        const float *rows = init + ((size_t)(init_per_offspring ? row : 0) * E + e0) * EnvT::INIT_W;
        double *ro = ep_return ? ep_return + (size_t)row * E + e0 : nullptr;
        int32_t *so = ep_steps ? ep_steps + (size_t)row * E + e0 : nullptr;
        gru_mfma4_batch<EnvT, FIXED_LENGTH>(tanh_tab, lds, net, lane, nb, rows, max_step, obs_mask, ro, so, valid);
    }
}

// ------------------------------------------------------------------------------------------------
// Lockstep GRU rollout with G offspring per wave, for envs whose step dwarfs the policy (the Box2D-style lander: ~20 000
// instructions per world step against ~600 for the GRU step of five episodes).  An env step costs a wave the same
// number of issue slots whether 5 or 40 of its lanes carry a live env, so the wave takes the envs of G offspring:
// lane l owns env (offspring (l >> 3) % G, episode l & 7), one call of the env step serves G x E envs.  The policy is
// evaluated offspring after offspring by all 64 lanes as in the one-offspring kernel (same arithmetic, same lane roles);
// the weights cannot all stay in registers, so each offspring's slice (~110 floats per lane) is re-read from its theta
// row (L2) before its GRU step -- ~27 KB per offspring and time step, 11 GB per C3 generation, next to 20 000
// instructions of solver per step.  Hidden states stay in registers (G x NP) and in the per-offspring LDS block.
// Measured (C3, 4096 offspring x 5 episodes): see DESIGN.md section 4.
template <typename EnvT, int G, int NP, bool ODD>
__device__ __forceinline__ void gru_lockstep_multi_batch(const TanhEntry *tanh_tab, GruLockstepLds<EnvT::S, EnvT::A> *lds,
                                                         const float *__restrict__ theta, int P, int row0, int n_rows,
                                                         int lane, int nb, const float *__restrict__ init,
                                                         int init_per_offspring, int E, int max_step, uint32_t obs_mask,
                                                         double *__restrict__ ep_return, int32_t *__restrict__ ep_steps)
{
    constexpr int S = EnvT::S, A = EnvT::A;
    const int slot = lane & 7, rep = lane >> 3, gl = rep % G;
    const int my_row_raw = row0 + gl;
    const bool row_valid = my_row_raw < n_rows;
    const int my_row = row_valid ? my_row_raw : n_rows - 1;
    const bool owner_valid = slot < nb && row_valid;
    typename EnvT::State st;
    EnvT::reset(st, init + ((size_t)(init_per_offspring ? my_row : 0) * E + (slot < nb ? slot : 0)) * EnvT::INIT_W, gl * 8 + slot);
    float hreg[G][NP];
#pragma unroll
    for (int g = 0; g < G; ++g) {
#pragma unroll
        for (int p = 0; p < NP; ++p) hreg[g][p] = 0.0f;                           // GymEnvModel.reset()
        const int row_g = row0 + g < n_rows ? row0 + g : n_rows - 1;
        GruLockstep<S, A> net;
        net.template load<true>(theta + (size_t)row_g * P, lane, lds[g]);         // W2 / b2 -> LDS (the registers are dropped)
        if (lane < 32) {
#pragma unroll
            for (int e = 0; e < GL_EB; ++e) lds[g].ah[e][lane][1] = 0.0f;
        }
    }
    wave_lds_sync();
#ifdef SES_PHASE_TIMERS
    phase_mark(-1);
#endif
    double ret = 0.0;
    int steps = 0;
    bool alive = true;
    for (int t = 0; t < max_step; ++t) {
        if (__ballot(alive & owner_valid) == 0ull) break;
#ifdef SES_PHASE_TIMERS
        phase_mark(10);
#endif
        float obs[S];
        EnvT::observe(st, obs);
        float logits[A];
#pragma unroll
        for (int o = 0; o < A; ++o) logits[o] = 0.0f;
#pragma unroll
        for (int g = 0; g < G; ++g) {
            // an offspring all of whose episodes are over needs no policy step any more (wave-uniform test): in the last
            // two thirds of a C3 rollout most waves carry ONE offspring with a long episode, and the other one's weight
            // re-read + GRU step was a tenth of their step
            if (__ballot(alive & owner_valid & (gl == g)) == 0ull) continue;
            const int row_g = row0 + g < n_rows ? row0 + g : n_rows - 1;
            GruLockstep<S, A> net;
            net.template load<false>(theta + (size_t)row_g * P, lane, lds[g]);
            if (rep == g) {                                                       // the first replica group of offspring g
#pragma unroll
                for (int k = 0; k < S; ++k) lds[g].obs[slot][k] = ((obs_mask >> k) & 1u) ? 0.0f : obs[k];
            }
            wave_lds_sync();
            net.template step<NP, ODD>(tanh_tab, lds[g], hreg[g], lane);
            float lg[A];
            net.logits_of(lds[g], lane, lg);
#pragma unroll
            for (int o = 0; o < A; ++o) logits[o] = gl == g ? lg[o] : logits[o];
        }
#ifdef SES_PHASE_TIMERS
#pragma unroll
        for (int o = 0; o < A; ++o) asm volatile("" : "+v"(logits[o]));
        phase_mark(11);
#endif
        bool term;
        const bool freeze = !(alive & owner_valid);
        const float r = EnvT::step(st, logits, tanh_tab, freeze, term);
        const int nsteps = steps + 1;
        const bool finished = term | (nsteps >= max_step);
        ret = alive ? ret + (double)r : ret;
        steps = alive ? nsteps : steps;
        alive = alive & !finished;
    }
    if (owner_valid && rep < G) {
        if (ep_return) ep_return[(size_t)my_row * E + slot] = ret;
        if (ep_steps) ep_steps[(size_t)my_row * E + slot] = steps;
    }
#ifdef SES_PHASE_TIMERS
    phase_flush();
#endif
}

template <typename EnvT, int G>
__global__ __launch_bounds__(256, 2) void k_rollout_gru_lockstep_multi(const float *__restrict__ theta,
                                                                    const float *__restrict__ init, int init_per_offspring,
                                                                    int n_rows, int E, int P, int max_step,
                                                                    uint32_t obs_mask, double *__restrict__ ep_return,
                                                                    int32_t *__restrict__ ep_steps)
{
    __shared__ TanhEntry tanh_tab[SES_TANH_N];
    __shared__ __attribute__((aligned(16))) GruLockstepLds<EnvT::S, EnvT::A> ldsv[4][G];
    stage_tanh_table(tanh_tab);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int row0 = (blockIdx.x * 4 + wave) * G;
    if (row0 >= n_rows) return;                       // (no workgroup-level synchronisation after the table staging)
#define SES_LSM_CASE(NP_, ODD_)                                                                                          \
    gru_lockstep_multi_batch<EnvT, G, NP_, ODD_>(tanh_tab, ldsv[wave], theta, P, row0, n_rows, lane, E, init,              \
                                                 init_per_offspring, E, max_step, obs_mask, ep_return, ep_steps)
    switch (E) {                                      // E <= GL_EB (the launcher checks)
        case 1: SES_LSM_CASE(1, true); break;
        case 2: SES_LSM_CASE(1, false); break;
        case 3: SES_LSM_CASE(2, true); break;
        case 4: SES_LSM_CASE(2, false); break;
        case 5: SES_LSM_CASE(3, true); break;
        case 6: SES_LSM_CASE(3, false); break;
        case 7: SES_LSM_CASE(4, true); break;
        default: SES_LSM_CASE(4, false); break;
    }
#undef SES_LSM_CASE
}

// ------------------------------------------------------------------------------------------------
// MFMA GRU rollout (ses_gru_mfma.h): one offspring per wave, up to 16 episodes are the columns of the
// v_mfma_f32_16x16x4_f32 tiles; lane l simulates the env of episode (l & 15) (four identical replicas, so no lane
// diverges).  Used for eval_ep_num >= 12.  launch_bounds(256, 2): 256 registers, two waves per SIMD.
// SQ counters at E = 16 (tools/prof_mfma.sh): 106 MFMAs + ~600 VALU instructions per step; the matrix pipe is busy 56 %
// of the kernel (MfmaUtil) and SQ_WAIT_INST_ANY is 54 % of the wave cycles: fp32 MFMA executes on the vector lanes
// (its peak IS the vector peak), so the contraction and the VALU phases of the two resident waves take turns instead
// of overlapping -- staggering the waves by a VALU phase changed nothing.
template <typename EnvT, bool FIXED_LENGTH>
__device__ __forceinline__ void gru_mfma_batch(const TanhEntry *tanh_tab, GruMfmaLds<EnvT::S, EnvT::A> &lds,
                                               const GruMfma<EnvT::S, EnvT::A> &net, int lane, int nb,
                                               const float *__restrict__ init_rows, int max_step, uint32_t obs_mask,
                                               double *__restrict__ ret_out, int32_t *__restrict__ steps_out,
                                               bool valid_row)
{
    constexpr int S = EnvT::S, A = EnvT::A;
    const int slot = lane & 15;
    const bool owner_valid = slot < nb;
    typename EnvT::State st;
    EnvT::reset(st, init_rows + (size_t)(owner_valid ? slot : 0) * EnvT::INIT_W, slot);   // padding columns replay episode 0
    float hreg[2][4];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) hreg[t][r] = 0.0f;                           // GymEnvModel.reset()
    wave_lds_sync();
    for (int i = lane; i < 32 * GM_EB; i += 64) (&lds.hT[0][0])[i] = 0.0f;
    double ret = 0.0;
    int steps = 0;
    bool alive = true;
    for (int t = 0; t < max_step; ++t) {
        if constexpr (!FIXED_LENGTH) {
            if (__ballot(alive & owner_valid) == 0ull) break;
        }
        float obs[S];
        EnvT::observe(st, obs);
        if (lane < GM_EB) {
#pragma unroll
            for (int k = 0; k < S; ++k) lds.obsT[k][lane] = ((obs_mask >> k) & 1u) ? 0.0f : obs[k];
        }
        wave_lds_sync();
        net.step(tanh_tab, lds, hreg, lane);
        const float4 lg4 = *reinterpret_cast<const float4 *>(&lds.logit[slot][0]);
        const float all[4] = {lg4.x, lg4.y, lg4.z, lg4.w};
        float logits[A];
#pragma unroll
        for (int o = 0; o < A; ++o) logits[o] = all[o];
        bool term;
        const bool freeze = FIXED_LENGTH ? false : !alive;
        const float r = EnvT::step(st, logits, tanh_tab, freeze, term);
        const int nsteps = steps + 1;
        const bool finished = term | (nsteps >= max_step);
        ret = alive ? ret + (double)r : ret;
        steps = alive ? nsteps : steps;
        alive = alive & !finished;
    }
    if (valid_row && lane < GM_EB && owner_valid) {
        if (ret_out) ret_out[slot] = ret;
        if (steps_out) steps_out[slot] = steps;
    }
}

template <typename EnvT, bool FIXED_LENGTH>
__global__ __launch_bounds__(256, 2) void k_rollout_gru_mfma(const float *__restrict__ theta,
                                                          const float *__restrict__ init, int init_per_offspring,
                                                          int n_rows, int E, int P, int max_step, uint32_t obs_mask,
                                                          double *__restrict__ ep_return,
                                                          int32_t *__restrict__ ep_steps)
{
    __shared__ TanhEntry tanh_tab[SES_TANH_N];
    __shared__ __attribute__((aligned(16))) GruMfmaLds<EnvT::S, EnvT::A> ldsv[4];
    stage_tanh_table(tanh_tab);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    int row = blockIdx.x * 4 + wave;
    const bool valid = row < n_rows;
    row = valid ? row : n_rows - 1;
    GruMfmaLds<EnvT::S, EnvT::A> &lds = ldsv[wave];
    GruMfma<EnvT::S, EnvT::A> net;
    net.load(theta + (size_t)row * P, lane, lds);
    wave_lds_sync();
    for (int e0 = 0; e0 < E; e0 += GM_EB) {
        const int nb = E - e0 < GM_EB ? E - e0 : GM_EB;
        const float *rows = init + ((size_t)(init_per_offspring ? row : 0) * E + e0) * EnvT::INIT_W;
        double *ro = ep_return ? ep_return + (size_t)row * E + e0 : nullptr;
        int32_t *so = ep_steps ? ep_steps + (size_t)row * E + e0 : nullptr;
        gru_mfma_batch<EnvT, FIXED_LENGTH>(tanh_tab, lds, net, lane, nb, rows, max_step, obs_mask, ro, so, valid);
    }
}

// LunarLanderContinuous-v2 (ses_lander.h): continuous control (tanh head, the env uses outputs 0 and 1 -- SURVEY 3.4-12), float
// rewards accumulated in float64 like the reference's python sum (loop.py:123).  GRU policy, one offspring (or one
// episode of it, ep_parallel) per wave, episodes one after the other; the MLP policies run in k_rollout_box2d_mlp.
__global__ __launch_bounds__(256, 2) void k_rollout_lander_gru(const float *__restrict__ theta,
                                                        const float *__restrict__ init, int init_per_offspring,
                                                        int n_rows, int E, int P, int max_step, uint32_t obs_mask,
                                                        double *__restrict__ ep_return, int32_t *__restrict__ ep_steps,
                                                        int ep_parallel)
{
    constexpr int S = 8, A = 4;
    __shared__ TanhEntry tanh_tab[SES_TANH_N];
    __shared__ __attribute__((aligned(16))) float vecs[4][64];
    __shared__ float terrain[4][LL_TERRAIN_ROW];                  // one terrain row per wave
    stage_tanh_table(tanh_tab);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    {
        const int unit = blockIdx.x * 4 + wave, n_units = ep_parallel ? n_rows * E : n_rows;   // see k_rollout_cartpole_gru
        const bool valid = unit < n_units;
        const int u = valid ? unit : n_units - 1;
        const int row = ep_parallel ? u / E : u;
        const int ep_begin = ep_parallel ? u - row * E : 0, ep_end = ep_parallel ? ep_begin + 1 : E;
        GruSlice<S, A> net;
        net.load(theta + (size_t)row * P, lane);
        float *vec = vecs[wave];
        for (int ep = ep_begin; ep < ep_end; ++ep) {
            LanderState st;
            ll_reset(st, init + ((size_t)(init_per_offspring ? row : 0) * E + ep) * 16, terrain[wave]);
            float h = 0.0f;
            wave_lds_sync();
            if (lane < 32) vec[2 * lane + 1] = 0.0f;
            wave_lds_sync();
            double ret = 0.0;
            int steps = 0;
            bool done = false;
            while (steps < max_step) {
                if (__builtin_amdgcn_readfirstlane((int)done)) break;
                float obs[S], logits[A];
                ll_obs(st, obs);
#pragma unroll
                for (int k = 0; k < S; ++k) obs[k] = ((obs_mask >> k) & 1u) ? 0.0f : obs[k];
                net.forward(tanh_tab, obs, h, vec, lane, logits);
                const float a0 = tanh_(tanh_tab, logits[0]), a1 = tanh_(tanh_tab, logits[1]);
                ret += (double)ll_step(st, a0, a1, done);
                steps += 1;
            }
            if (valid && lane == 0) {
                ep_return[(size_t)row * E + ep] = ret;
                if (ep_steps) ep_steps[(size_t)row * E + ep] = steps;
            }
        }
    }
}

// MLP policies on the Box2D-style envs (LunarLander: conf/lunarlander.yaml, BipedalWalker: conf/bipedalwalker.yaml):
// LPE lanes per env (LPE <= 8: 32 / LPE hidden units each for the forward; above: the forward on every 16-lane row), every
// lane of a group carries the env.  A wave-step costs a + b x (different envs in the wave) with a large a -- the solver's
// sequential iterations, whatever the number of lanes that carry an env -- so box2d_wave_shape() spreads the population
// over every wave slot of the chip with as few envs per wave as that allows (envs_per_wave), LPE follows from it, and
// the lane groups past the last env shadow it (same env, same path).
// Single-wave workgroups (they spread over all SIMDs and retire independently).  EnvB adapts an env.
struct LanderMlpEnv {
    static constexpr int S = 8, A = 4, INIT_W = 16, ROW = LL_TERRAIN_ROW;
    static constexpr int WAVES_PER_SIMD = 2;                      // 256 registers for the kernel and for ll_step: enough
    using State = LanderState;
    __device__ static __forceinline__ void reset(State &s, const float *__restrict__ u, float *row) { ll_reset(s, u, row); }
    __device__ static __forceinline__ void observe(const State &s, float (&obs)[S]) { ll_obs(s, obs); }
    __device__ static __forceinline__ float step(State &s, const float (&act)[A], bool &done) { return ll_step(s, act[0], act[1], done); }
};

struct WalkerMlpEnv {
    static constexpr int S = 24, A = 4, INIT_W = 4, ROW = BW_TERRAIN_ROW;
    // one wave per SIMD: bw_step (compiled for its callers' budget) may then use all 512 registers -- the walker's
    // world does not fit 256, and what spills goes to AGPRs (one v_accvgpr move) instead of scratch memory (a trip
    // to L2 or HBM inside the 180-iteration solver loop)
    static constexpr int WAVES_PER_SIMD = 1;
    using State = WalkerState;
    __device__ static __forceinline__ void reset(State &s, const float *__restrict__ u, float *row) { bw_reset(s, u, row); }
    __device__ static __forceinline__ void observe(const State &s, float (&obs)[S]) { bw_obs(s, obs); }
    __device__ static __forceinline__ float step(State &s, const float (&act)[A], bool &done)
    {
        return bw_step(s, act[0], act[1], act[2], act[3], done);
    }
};

template <class EnvB, int LPE>
__global__ __launch_bounds__(64, EnvB::WAVES_PER_SIMD) void k_rollout_box2d_mlp(const float *__restrict__ theta,
                                                           const float *__restrict__ init, int init_per_offspring,
                                                           int n_rows, int E, int P, int max_step, uint32_t obs_mask,
                                                           int envs_per_wave, double *__restrict__ ep_return,
                                                           int32_t *__restrict__ ep_steps)
{
    constexpr int S = EnvB::S, A = EnvB::A;
    __shared__ TanhEntry tanh_tab[SES_TANH_N];
    __shared__ float terrain[64 / LPE][EnvB::ROW];                // one terrain row per env
    stage_tanh_table(tanh_tab);
    // envs_per_wave <= 64 / LPE envs in this wave; the lane groups past the last one shadow it (same env, same path:
    // they add nothing to what the wave executes) -- a wave-step costs about as much as it carries DIFFERENT envs
    const int n_env = n_rows * E;
    const int group = (int)threadIdx.x / LPE;
    const int slot = group < envs_per_wave ? group : envs_per_wave - 1;
    int env = (int)blockIdx.x * envs_per_wave + slot;
    const int sub = (int)(threadIdx.x % LPE);
    const bool valid = env < n_env && group < envs_per_wave;
    env = env < n_env ? env : n_env - 1;
    const int row = env / E, ep = env - row * E;
    typename EnvB::State st;
    EnvB::reset(st, init + ((size_t)(init_per_offspring ? row : 0) * E + ep) * EnvB::INIT_W, terrain[slot]);
#ifdef SES_PHASE_TIMERS
    phase_mark(-1);
#endif
    double ret = 0.0;
    int steps = 0;
    bool done = false;
    for (int t = 0; t < max_step; ++t) {
        if (__ballot(!done) == 0ull) break;
        // the lane's weight slice is re-read from the (L2-resident) row every step: a few dozen loads next to a
        // 20 000-instruction world step, and nothing of the policy has to stay in registers across it
        float obs[S], logits[A], act[A];
#ifdef SES_PHASE_TIMERS
        phase_mark(10);
#endif
        EnvB::observe(st, obs);
#pragma unroll
        for (int k = 0; k < S; ++k) obs[k] = ((obs_mask >> k) & 1u) ? 0.0f : obs[k];
        if constexpr (LPE > 8) {
            // 16, 32 or 64 lanes per env: small populations, where an env that has a wave (or a good part of one) to itself
            // pays for its own contacts, impacts and position iterations only, not for the union over its wave-mates'.
            // The policy runs on each 16-lane row of the env's lanes (2 hidden units per lane, MlpSlice<.., 16>: ~90
            // instructions and 30 weight loads per step; round 2 had every lane evaluate the whole policy from streamed
            // weights: ~930 instructions and 420 dependent-ish loads, a tenth of a lander step)
            MlpSlice<S, A, 16> net;
            net.load(theta + (size_t)row * P, (int)(threadIdx.x & 15));
            net.forward(tanh_tab, obs, logits);
        } else if constexpr (LPE >= 4) {
            MlpSlice<S, A, LPE> net;
            net.load(theta + (size_t)row * P, sub);
            net.forward(tanh_tab, obs, logits);
        } else {
            mlp_forward_streamed<S, A, LPE>(theta + (size_t)row * P, sub, tanh_tab, obs, logits);
        }
#pragma unroll
        for (int k = 0; k < A; ++k) act[k] = tanh_(tanh_tab, logits[k]);
#ifdef SES_PHASE_TIMERS
#pragma unroll
        for (int k = 0; k < A; ++k) asm volatile("" : "+v"(act[k]));   // the policy is done before the mark
        phase_mark(11);
#endif
        if (!done) {                                               // a finished env is frozen
            ret += (double)EnvB::step(st, act, done);
            steps += 1;
        }
    }
    if (valid && sub == 0) {
        ep_return[env] = ret;
        if (ep_steps) ep_steps[env] = steps;
    }
#ifdef SES_PHASE_TIMERS
    phase_flush();
#endif
}

// simple_spread: NA agents per env share the offspring's MLP (utils.py:4-8: one deepcopy per agent, same
// weights); per cycle every agent is evaluated on its own observation, then the world steps once
// (pettingzoo_wrapper.py:36-52).  8 lanes per env: with S = 6*NA inputs the lane's weight slice is
// 4 x (S + 1 + 5) registers.  Episodes last max_cycles (25) steps, so this kernel is short and dominated by
// the one-time weight load; nothing is early-exited (all agents finish together).
template <int NA>
__global__ __launch_bounds__(64) void k_rollout_spread_mlp(const float *__restrict__ theta,
                                                           const float *__restrict__ init, int init_per_offspring,
                                                           int n_rows, int E, int P, int max_cycles,
                                                           double *__restrict__ ep_return)
{
    constexpr int LPE = 8, S = 6 * NA, A = 5;
    __shared__ TanhEntry tanh_tab[SES_TANH_N];
    stage_tanh_table(tanh_tab);
    const long long gtid = (long long)blockIdx.x * 64 + threadIdx.x;
    const int n_env = n_rows * E;
    int env = (int)(gtid / LPE);
    const int sub = (int)(threadIdx.x % LPE);
    const bool valid = env < n_env;
    env = valid ? env : n_env - 1;
    const int row = env / E;
    const int ep = env - row * E;
    MlpSlice<S, A, LPE> net;
    net.load(theta + (size_t)row * P, sub);
    const float *s0 = init + ((size_t)(init_per_offspring ? row : 0) * E + ep) * (4 * NA);
    SpreadState<NA> st;
#pragma unroll
    for (int i = 0; i < NA; ++i) {
        st.ax[i] = s0[2 * i]; st.ay[i] = s0[2 * i + 1];
        st.vx[i] = 0.0f; st.vy[i] = 0.0f;
        st.lx[i] = s0[2 * NA + 2 * i]; st.ly[i] = s0[2 * NA + 2 * i + 1];
    }
    double ret = 0.0;
    for (int t = 0; t < max_cycles; ++t) {
        int action[NA];
#pragma unroll
        for (int a = 0; a < NA; ++a) {
            float obs[S], logits[A];
            spread_obs<NA>(st, a, obs);
            net.forward(tanh_tab, obs, logits);
            action[a] = argmax_first<A>(logits);
        }
        ret += (double)spread_step<NA, true>(st, action, sub & 3);
    }
    if (valid && sub == 0) ep_return[env] = ret;
}

__global__ void k_fitness_mean(const double *__restrict__ ep_return, int n_rows, int E, float *__restrict__ fitness,
                               unsigned long long *__restrict__ stamp)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (stamp && i == 0) *stamp = real_time();                          // end of the rollout phase (ses_set_stamp)
    if (i >= n_rows) return;
    double total = 0.0;
    for (int e = 0; e < E; ++e) total += ep_return[(size_t)i * E + e];
    fitness[i] = (float)(total / (double)E);
}

// The same mean, and the fitness exchange of a sharded run in the same launch (ses_run_generations, "fused_fitness_exchange"):
// every thread also stores its value as an 8-byte {exchange number, value} granule into the mailbox of EVERY rank (its own
// included) -- one store per rank, the data is its own flag -- where the ranks' rank kernels poll the tiles they sort
// (k_rank_sort_search).  No exchange launch between the rollout and the tail.
__global__ void k_fitness_mean_granules(const double *__restrict__ ep_return, int n_rows, int E, float *__restrict__ fitness,
                                        unsigned long long *__restrict__ stamp, P2pGranuleView gv)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (stamp && i == 0) *stamp = real_time();
    if (i >= n_rows) return;
    double total = 0.0;
    for (int e = 0; e < E; ++e) total += ep_return[(size_t)i * E + e];
    const float f = (float)(total / (double)E);
    fitness[i] = f;
    for (int r = 0; r < gv.world; ++r) granule_store(gv.dst[r] + i, gv.seq, f2u(f));
}

// ------------------------------------------------------------------------------------------------
// Standalone SoA env step: pure streaming, 16 B per lane per array (7 loads + 6 stores = 52 B/env).
template <bool FIXED_LENGTH>
__device__ __forceinline__ void env_step_one(float &x, float &xd, float &th, float &thd, int action, float &ret,
                                             uint32_t &status, int max_step)
{
    const bool done = (status >> 31) != 0u;
    const uint32_t steps = status & 0x7fffffffu;
    CartPoleState s{x, xd, th, thd};
    const bool term = cartpole_step(s, action);
    const bool advance = FIXED_LENGTH ? true : !done;
    x = advance ? s.x : x;
    xd = advance ? s.xd : xd;
    th = advance ? s.th : th;
    thd = advance ? s.thd : thd;
    const uint32_t nsteps = steps + 1u;
    const bool now_done = term | (max_step > 0 && (int)nsteps >= max_step);
    ret = done ? ret : ret + 1.0f;
    status = done ? status : (nsteps | ((uint32_t)now_done << 31));
}

// 16-byte non-temporal accesses: the state is streamed once per step and never re-read by this kernel,
// so it should not displace anything in L2 / Infinity Cache (measured +7 % over plain loads/stores).
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef int32_t i32x4 __attribute__((ext_vector_type(4)));

template <bool FIXED_LENGTH>
__global__ __launch_bounds__(256) void k_env_step_cartpole_v4(int n4, int max_step, f32x4 *__restrict__ x,
                                                              f32x4 *__restrict__ xd, f32x4 *__restrict__ th,
                                                              f32x4 *__restrict__ thd,
                                                              const i32x4 *__restrict__ action,
                                                              f32x4 *__restrict__ ret, u32x4 *__restrict__ status)
{
    const int stride = gridDim.x * blockDim.x;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
        f32x4 vx = __builtin_nontemporal_load(x + i), vxd = __builtin_nontemporal_load(xd + i),
              vth = __builtin_nontemporal_load(th + i), vthd = __builtin_nontemporal_load(thd + i),
              vr = __builtin_nontemporal_load(ret + i);
        const i32x4 va = __builtin_nontemporal_load(action + i);
        u32x4 vs = __builtin_nontemporal_load(status + i);
#pragma unroll
        for (int l = 0; l < 4; ++l) {
            float ex = vx[l], exd = vxd[l], eth = vth[l], ethd = vthd[l], er = vr[l];
            uint32_t es = vs[l];
            env_step_one<FIXED_LENGTH>(ex, exd, eth, ethd, va[l], er, es, max_step);
            vx[l] = ex; vxd[l] = exd; vth[l] = eth; vthd[l] = ethd; vr[l] = er; vs[l] = es;
        }
        __builtin_nontemporal_store(vx, x + i);
        __builtin_nontemporal_store(vxd, xd + i);
        __builtin_nontemporal_store(vth, th + i);
        __builtin_nontemporal_store(vthd, thd + i);
        __builtin_nontemporal_store(vr, ret + i);
        __builtin_nontemporal_store(vs, status + i);
    }
}

// The env-step kernel's 13 streams with no arithmetic in between (7 x 16-B non-temporal loads, 6 x 16-B non-temporal
// stores per lane, same grid): what the memory system of THIS box gives this access pattern -- the ceiling bench.py
// prints next to the env-step kernel (ses_stream_probe).  Every loaded vector passes through an empty asm statement, so
// that neither a load nor a store-back of an unchanged value can be dropped (tests/test_profiles_current.py counts them).
__global__ __launch_bounds__(256) void k_stream_probe13(int n4, f32x4 *__restrict__ x, f32x4 *__restrict__ xd,
                                                        f32x4 *__restrict__ th, f32x4 *__restrict__ thd,
                                                        const i32x4 *__restrict__ action, f32x4 *__restrict__ ret,
                                                        u32x4 *__restrict__ status)
{
    const int stride = gridDim.x * blockDim.x;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
        f32x4 vx = __builtin_nontemporal_load(x + i);
        const f32x4 vxd = __builtin_nontemporal_load(xd + i), vth = __builtin_nontemporal_load(th + i),
                    vthd = __builtin_nontemporal_load(thd + i), vr = __builtin_nontemporal_load(ret + i);
        const i32x4 va = __builtin_nontemporal_load(action + i);
        const u32x4 vs = __builtin_nontemporal_load(status + i);
        asm volatile("" ::"v"(va));                               // the action stream is loaded (and consumed here), never stored
        // opaque to the optimiser: without this, storing a value back to the address it was loaded from is dropped
        f32x4 wxd = vxd, wth = vth, wthd = vthd, wr = vr;
        u32x4 ws = vs;
        asm volatile("" : "+v"(vx), "+v"(wxd), "+v"(wth), "+v"(wthd), "+v"(wr), "+v"(ws));
        __builtin_nontemporal_store(vx, x + i);
        __builtin_nontemporal_store(wxd, xd + i);
        __builtin_nontemporal_store(wth, th + i);
        __builtin_nontemporal_store(wthd, thd + i);
        __builtin_nontemporal_store(wr, ret + i);
        __builtin_nontemporal_store(ws, status + i);
    }
}

template <bool FIXED_LENGTH>
__global__ void k_env_step_cartpole_scalar(int first, int n, int max_step, float *x, float *xd, float *th, float *thd,
                                           const int32_t *action, float *ret, uint32_t *status)
{
    const int i = first + blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float vx = x[i], vxd = xd[i], vth = th[i], vthd = thd[i], vr = ret[i];
    uint32_t vs = status[i];
    env_step_one<FIXED_LENGTH>(vx, vxd, vth, vthd, action[i], vr, vs, max_step);
    x[i] = vx; xd[i] = vxd; th[i] = vth; thd[i] = vthd; ret[i] = vr; status[i] = vs;
}

// ------------------------------------------------------------------------------------------------
// Standalone MLP forward, 4 lanes per (row, obs) pair.
template <int S, int A>
__global__ __launch_bounds__(64) void k_policy_forward_mlp(const float *__restrict__ theta,
                                                           const float *__restrict__ obs_in, int n, int P,
                                                           float *__restrict__ logits_out, float *__restrict__ act_out,
                                                           int32_t *__restrict__ action_out)
{
    constexpr int LPE = 4;
    __shared__ TanhEntry tanh_tab[SES_TANH_N];
    stage_tanh_table(tanh_tab);
    const long long gtid = (long long)blockIdx.x * 64 + threadIdx.x;
    int i = (int)(gtid / LPE);
    const int sub = (int)(threadIdx.x % LPE);
    const bool valid = i < n;
    i = valid ? i : n - 1;
    MlpSlice<S, A, LPE> net;
    net.load(theta + (size_t)i * P, sub);
    float obs[S];
#pragma unroll
    for (int k = 0; k < S; ++k) obs[k] = obs_in[(size_t)i * S + k];
    float logits[A];
    net.forward(tanh_tab, obs, logits);
    const int action = argmax_first<A>(logits);
    if (valid && sub == 0) {
#pragma unroll
        for (int k = 0; k < A; ++k) {
            logits_out[(size_t)i * A + k] = logits[k];
            if (act_out) act_out[(size_t)i * A + k] = tanh_(tanh_tab, logits[k]);
        }
        action_out[i] = action;
    }
}

// Standalone GRU forward: one (row, obs, hidden) triple per wavefront.
template <int S, int A>
__global__ __launch_bounds__(256) void k_policy_forward_gru(const float *__restrict__ theta,
                                                            const float *__restrict__ obs_in,
                                                            float *__restrict__ hidden, int n, int P,
                                                            float *__restrict__ logits_out, float *__restrict__ act_out,
                                                            int32_t *__restrict__ action_out)
{
    __shared__ TanhEntry tanh_tab[SES_TANH_N];
    __shared__ __attribute__((aligned(16))) float vecs[4][64];
    stage_tanh_table(tanh_tab);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    int i = blockIdx.x * 4 + wave;
    const bool valid = i < n;
    i = valid ? i : n - 1;
    GruSlice<S, A> net;
    net.load(theta + (size_t)i * P, lane);
    float *vec = vecs[wave];
    float obs[S];
#pragma unroll
    for (int k = 0; k < S; ++k) obs[k] = obs_in[(size_t)i * S + k];
    float h = hidden[(size_t)i * H + (lane & 31)];
    if (lane < 32) vec[2 * lane + 1] = h;
    wave_lds_sync();
    float logits[A];
    net.forward(tanh_tab, obs, h, vec, lane, logits);
    const int action = argmax_first<A>(logits);
    if (valid && lane < 32) hidden[(size_t)i * H + lane] = h;
    if (valid && lane == 0) {
#pragma unroll
        for (int k = 0; k < A; ++k) {
            logits_out[(size_t)i * A + k] = logits[k];
            if (act_out) act_out[(size_t)i * A + k] = tanh_(tanh_tab, logits[k]);
        }
        action_out[i] = action;
    }
}

static int pick_lanes_per_env(const ses_handle *h, long long n_env)
{
    if (h->cfg.lanes_per_env) return h->cfg.lanes_per_env;
    // Measured on MI355X (round 1 sweep, profiles/r01_rollout_occupancy_sweep.txt): 4 lanes per env wins from 20 480 envs (1280 waves) up to
    // 327 680 envs; below ~10 000 envs only LPE = 8 still gives every SIMD a wavefront, and while even that leaves the
    // population within one wave per SIMD (4096 envs) 16 lanes per env make the lone wave's step shorter still
    // (83 instead of 104 instructions: conf/cartpole.yaml's 480 envs are 500 sequential steps of such a wave).
    if (n_env * 4 / 64 >= 640) return 4;
    return n_env <= 4096 ? 16 : 8;
}

template <int LPE, int BLOCK, bool PK = false>
static void launch_rollout_b(const ses_handle *h, const float *theta, const float *init, int per, int n_rows, int mode,
                             double *ep_return, int32_t *ep_steps)
{
    const long long threads = (long long)n_rows * h->cfg.eval_ep_num * LPE;
    const int blocks = ceil_div(threads, BLOCK);
    if (mode == SES_MODE_FIXED_LENGTH)
        hipLaunchKernelGGL((k_rollout_cartpole_mlp<LPE, true, BLOCK, false, PK>), dim3(blocks), dim3(BLOCK), 0, h->stream, theta,
                           init, per, n_rows, h->cfg.eval_ep_num, h->P, h->cfg.max_step, h->obs_mask, ep_return,
                           ep_steps);
    else
        hipLaunchKernelGGL((k_rollout_cartpole_mlp<LPE, false, BLOCK, false, PK>), dim3(blocks), dim3(BLOCK), 0, h->stream, theta,
                           init, per, n_rows, h->cfg.eval_ep_num, h->P, h->cfg.max_step, h->obs_mask, ep_return,
                           ep_steps);
}

// The packed step (ses_policy_pk.h) pays where a wave has its SIMD to itself: populations of at most one wave per SIMD at the
// chosen lanes per env.  ses_set_tuning "rollout_packed": -1 = this rule (default), 0 = never, 1 = whenever the split is 8 or 16.
static bool cartpole_mlp_packed(const ses_handle *h, int lpe, long long episodes)
{
    if (lpe != 8 && lpe != 16) return false;
    if (h->tune_rollout_packed >= 0) return h->tune_rollout_packed != 0;
    return ceil_div(episodes * lpe, 64) <= h->tune_rollout_waves8;          // (the knob holds the chip's SIMD count: 1024)
}

template <int LPE>
static void launch_rollout(const ses_handle *h, const float *theta, const float *init, int per, int n_rows, int mode,
                           double *ep_return, int32_t *ep_steps)
{
    if constexpr (LPE == 8 || LPE == 16) {
        if (h->tune_rollout_block != 256 && cartpole_mlp_packed(h, LPE, (long long)n_rows * h->cfg.eval_ep_num)) {
            launch_rollout_b<LPE, 64, true>(h, theta, init, per, n_rows, mode, ep_return, ep_steps);
            return;
        }
    }
    if (h->tune_rollout_block == 256) launch_rollout_b<LPE, 256>(h, theta, init, per, n_rows, mode, ep_return, ep_steps);
    else launch_rollout_b<LPE, 64>(h, theta, init, per, n_rows, mode, ep_return, ep_steps);
}

// eval_ep_num from which the GRU rollouts run on the matrix cores (ses_set_tuning "gru_mfma_min_e", default 12).
// Measured, POMDP CartPole, 4096 offspring x 500 steps: the MFMA form takes 5.1 ms for any E <= 16 (the padded tile
// costs the same), the VALU lockstep form 2.4 / 3.5 / 5.6 / 7.2 ms at E = 5 / 8 / 12 / 16 -- the crossover is at 12.
// Envs per wave and lanes per env of the Box2D MLP rollout.  A wave-step costs about as much as the wave carries
// different envs (the union of their contact rows, impacts, position iterations), so the population is spread over every
// wave slot of the chip -- 1024 SIMDs x EnvB::WAVES_PER_SIMD, all resident at once -- with as few envs per wave as that
// allows, and the lanes per env are the largest power of two that fits them.  BipedalWalker, 4096 x 5 episodes: 32 envs
// on each of 640 waves 374 ms, 16 on each of 1280 (two rounds on 1024 slots) 356 ms, 20 on each of 1024 -> see DESIGN.md.
// ses_set_tuning "box2d_lanes_per_env" forces LPE (and 64 / LPE envs per wave unless "box2d_envs_per_wave" is set too).
static void box2d_wave_shape(const ses_handle *h, long long episodes, int waves_per_simd, int &lpe, int &epw)
{
    if (h->tune_box2d_lpe) {
        lpe = h->tune_box2d_lpe;
        epw = h->tune_box2d_epw ? h->tune_box2d_epw : 64 / lpe;
        if (epw > 64 / lpe) epw = 64 / lpe;
        return;
    }
    const long long slots = 1024ll * waves_per_simd;
    long long need = h->tune_box2d_epw ? h->tune_box2d_epw : (episodes + slots - 1) / slots;
    epw = (int)(need < 1 ? 1 : need > 64 ? 64 : need);
    lpe = 64;
    while (lpe > 1 && epw * lpe > 64) lpe >>= 1;
}

template <class EnvB>
static void launch_box2d_mlp(ses_handle *h, const float *theta, const float *init, int per, int n_rows, double *epr,
                             int32_t *ep_steps)
{
    const long long episodes = (long long)n_rows * h->cfg.eval_ep_num;
    int lpe, epw;
    box2d_wave_shape(h, episodes, EnvB::WAVES_PER_SIMD, lpe, epw);
    const dim3 grid(ceil_div(episodes, epw)), block(64);
#define SES_BOX2D_LAUNCH(L)                                                                                          \
    hipLaunchKernelGGL((k_rollout_box2d_mlp<EnvB, L>), grid, block, 0, h->stream, theta, init, per, n_rows,           \
                       h->cfg.eval_ep_num, h->P, h->cfg.max_step, h->obs_mask, epw, epr, ep_steps)
    if (lpe == 64) SES_BOX2D_LAUNCH(64);
    else if (lpe == 32) SES_BOX2D_LAUNCH(32);
    else if (lpe == 16) SES_BOX2D_LAUNCH(16);
    else if (lpe == 8) SES_BOX2D_LAUNCH(8);
    else if (lpe == 4) SES_BOX2D_LAUNCH(4);
    else if (lpe == 2) SES_BOX2D_LAUNCH(2);
    else SES_BOX2D_LAUNCH(1);
#undef SES_BOX2D_LAUNCH
}

static int gru_mfma_min_e(const ses_handle *h) { return h->tune_gru_mfma_min_e; }

// eval_ep_num for which the CartPole GRU rollout takes the 4x4x1 MFMA step (ses_set_tuning "gru_mfma4_min_e" ... 8; 0 = never)
static bool gru_mfma4(const ses_handle *h)
{
    return h->tune_gru_mfma4_min_e > 0 && h->cfg.eval_ep_num >= h->tune_gru_mfma4_min_e && h->cfg.eval_ep_num <= G4_EB;
}

// Small populations: the chip is far from full and what a rollout costs is the latency of max_step sequential env
// steps.  The lockstep kernel spends ~1.7 us per step (all E episodes of an offspring in one wave), the
// episode-after-episode kernel ~0.75 us per step and episode -- launched with one wave per (offspring, episode) it
// finishes in one episode's time.  Measured, POMDP CartPole, E = 5, 500 steps (lockstep / episode-parallel, ms):
// 96 offspring 0.87 / 0.37, 400: 0.87 / 0.46, 800: 0.87 / 0.82, 1200: 1.22 / 1.13, 1600: 1.23 / 1.48, 4096: 2.41 / 3.50.
// (ses_set_tuning "gru_ep_parallel_max": (offspring x episode) waves up to which the form is used, default 4096)
static bool gru_episode_parallel(const ses_handle *h, long long episodes)
{
    return episodes <= h->tune_gru_ep_parallel_max && h->cfg.eval_ep_num > 1;
}

// ses_set_tuning "gru_sequential" = 1 selects the episode-after-episode GRU kernels
static bool gru_sequential(const ses_handle *h) { return h->tune_gru_sequential != 0; }

// Offspring per wave of the lockstep lander rollout (k_rollout_gru_lockstep_multi): as many as leave about two waves
// per SIMD (2048 on the chip), all resident at once.  The env step is a sequential 20 000-instruction routine: a
// rollout costs (steps of the longest episode) x (time of one wave-step), a wave-step costs the same for 5 or 20 envs,
// and two waves on a SIMD fill each other's dependency stalls.  Measured, C3 (4096 offspring x 5 episodes, one box):
// 1 offspring per wave 44.8 ms, 2: 31.8 ms, 4: 35.9 ms (one wave per SIMD, and with 20 envs in a wave most steps have
// some env on the ground, i.e. run the contact rows).
// ses_set_tuning "lander_offspring_per_wave": 0 = this rule, 1 / 2 / 4 = forced.
static int lander_offspring_per_wave(const ses_handle *h, int n_rows)
{
    if (h->tune_lander_per_wave) return h->tune_lander_per_wave;
    // round 3: a finished offspring's GRU step is skipped, which makes four offspring per wave the better choice from
    // ~3000 offspring on (C3, 4096 x 5 x <= 300: 40.6 ms at two per wave, 37.4 at four, same box; tools/c3_breakdown.py)
    return n_rows >= 3072 ? 4 : (n_rows >= 1536 ? 2 : 1);
}

// Which split of the lanes runs a CartPole MLP population of `episodes` envs.  Every split evaluates the same canonical
// arithmetic; what differs is how much the busiest SIMD has to issue per env step and with how many waves it shares the
// issue port.  Model (profiles/r03_ab_mix_light.txt, MI355X): a wave's loop body is 160 / 104 / 84 VALU instructions at 4 / 8 / 16
// lanes per env; a SIMD that holds k waves issues one of their instructions every 5.0 / 3.45 / 3.1 / 2.95 cycles (k = 1,
// 2, 3, >= 4: a lone wave waits for its own dependences).  Candidates: the pure splits, and "one light wave per SIMD
// (8 or 16 lanes per env) + the rest at 4 lanes per env".  Measured against the model at 3072 / 4096 / 5120 offspring x 5
// episodes: pure 8 (167.7 us) / light 16 + 4 (195.0 us; round 2's light 8 + 4: 213.8) / pure 4 (239.5), all as predicted.
struct MlpSplit {
    int light;      // 0: pure split at `lpe` lanes per env; 8 / 16: mixed, the first waves at this many lanes per env
    int lpe;        // pure: lanes per env; mixed: lanes per env of the remaining waves (4, or 16 behind first waves of 8)
};

static MlpSplit choose_cartpole_mlp_split(const ses_handle *h, long long episodes)
{
    if (h->cfg.lanes_per_env) return MlpSplit{0, h->cfg.lanes_per_env};
    // round 6: 32 lanes per env (77 VALU instructions per step against 83 at 16, but two more dependent steps in fc2) -- a knob,
    // off by default: profiles/r06_small_populations.txt has the A/B that decides it
    if (h->tune_rollout_lpe32_max > 0 && episodes <= h->tune_rollout_lpe32_max) return MlpSplit{0, 32};
    if (episodes > 49152) return MlpSplit{0, 4};                       // large populations: 4 lanes per env (round 1 sweep)
    const int simds = h->tune_rollout_waves8;                           // 1024 = 256 CUs x 4 (knob: the first waves of a mix)
    auto instr = [](int lpe) { return lpe == 4 ? 160.0 : (lpe == 8 ? 104.0 : 84.0); };
    auto cadence = [](long long k) { return k <= 1 ? 5.0 : (k == 2 ? 3.45 : (k == 3 ? 3.1 : 2.95)); };
    MlpSplit best{0, 8};
    double best_cost = -1.0;
    auto consider = [&](MlpSplit sp, double cost) {
        if (best_cost < 0.0 || cost < best_cost) { best_cost = cost; best = sp; }
    };
    const bool mix_only = h->tune_rollout_mix_light != 0;
    if (!mix_only) {
        for (int lpe : {16, 8, 4}) {
            const long long k = ceil_div(ceil_div(episodes * lpe, 64), simds);
            consider(MlpSplit{0, lpe}, k * instr(lpe) * cadence(k));
        }
    }
    if ((h->tune_rollout_mix || mix_only) && episodes > 8192) {
        for (int light : {16, 8}) {
            if (mix_only && light != h->tune_rollout_mix_light) continue;
            const long long envs_light = (long long)simds * (64 / light);
            if (episodes <= envs_light) continue;
            const long long kh = ceil_div(ceil_div(episodes - envs_light, 16), simds);
            consider(MlpSplit{light, 4}, (instr(light) + 160.0 * kh) * cadence(1 + kh));
        }
        // 8 lanes per env on every SIMD, the rest at 16 lanes per env (round 6; knob "rollout_mix_8_16", default on)
        const long long envs8 = (long long)simds * 8;
        if (h->tune_rollout_mix_8_16 && !mix_only && episodes > envs8) {
            const long long k16 = ceil_div(ceil_div(episodes - envs8, 4), simds);
            consider(MlpSplit{8, 16}, (104.0 + 84.0 * k16) * cadence(1 + k16));
        }
    }
    return best;
}

static void launch_cartpole_mlp(const ses_handle *h, const float *theta, const float *init, int per, int n_rows,
                                int mode, double *epr, int32_t *ep_steps)
{
    const long long episodes = (long long)n_rows * h->cfg.eval_ep_num;
    const MlpSplit sp = choose_cartpole_mlp_split(h, episodes);
    if (sp.light) {
        // one wave of the first kind per SIMD (the dispatcher deals the first workgroups one per SIMD) + the rest
        const int epw = 64 / sp.light, knob = h->tune_rollout_waves8, epw_rest = 64 / sp.lpe;
        const int waves_light = (long long)knob * epw < episodes ? knob : (int)(episodes / epw);
        const int waves_rest = ceil_div(episodes - (long long)epw * waves_light, epw_rest);
        const dim3 grid(waves_light + waves_rest), block(64);
#define SES_MIX_LAUNCH(FIXED_, LIGHT_, REST_)                                                                            \
    hipLaunchKernelGGL((k_rollout_cartpole_mlp_mix<FIXED_, LIGHT_, REST_>), grid, block, 0, h->stream, theta, init, per,   \
                       n_rows, h->cfg.eval_ep_num, h->P, h->cfg.max_step, h->obs_mask, waves_light, epr, ep_steps)
        if (mode == SES_MODE_FIXED_LENGTH) {
            if (sp.lpe == 16) SES_MIX_LAUNCH(true, 8, 16);
            else if (sp.light == 16) SES_MIX_LAUNCH(true, 16, 4);
            else SES_MIX_LAUNCH(true, 8, 4);
        } else {
            if (sp.lpe == 16) SES_MIX_LAUNCH(false, 8, 16);
            else if (sp.light == 16) SES_MIX_LAUNCH(false, 16, 4);
            else SES_MIX_LAUNCH(false, 8, 4);
        }
#undef SES_MIX_LAUNCH
        return;
    }
    switch (sp.lpe) {
        case 1: launch_rollout<1>(h, theta, init, per, n_rows, mode, epr, ep_steps); break;
        case 2: launch_rollout<2>(h, theta, init, per, n_rows, mode, epr, ep_steps); break;
        case 4: launch_rollout<4>(h, theta, init, per, n_rows, mode, epr, ep_steps); break;
        case 16: launch_rollout<16>(h, theta, init, per, n_rows, mode, epr, ep_steps); break;
        case 32: launch_rollout<32>(h, theta, init, per, n_rows, mode, epr, ep_steps); break;
        default: launch_rollout<8>(h, theta, init, per, n_rows, mode, epr, ep_steps); break;
    }
}

// launch shape of the standalone env-step kernel (and of the probe that has to match it): see ses_env_step
struct EnvStepShape {
    int blocks, block, lds;
};
// The LDS reservation that limits the waves in flight.  Knob "env_step_lds_bytes" >= 0: taken as given.  -1 (default):
// derived from the device -- its LDS per CU divided by the workgroups per CU that make "env_step_waves_per_cu" (7) waves,
// rounded down to 256 bytes and checked against the occupancy calculator (the allocation granule is the hardware's
// business: if the rounded figure still lets one more workgroup in, or one fewer, it is moved by 256 bytes until the
// calculator agrees).  Resolved once per (block, knobs); env_step_wpc is what the calculator says in the end.
static void env_step_resolve(ses_handle *h)
{
    const int block = h->tune_env_step_block;
    if (h->env_step_key[0] == block && h->env_step_key[1] == h->tune_env_step_lds && h->env_step_key[2] == h->tune_env_step_waves) return;
    auto occupancy = [&](int lds) {
        int nb = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_env_step_cartpole_v4<true>, block, (size_t)lds) != hipSuccess) {
            (void)hipGetLastError();
            nb = 0;
        }
        return nb;
    };
    int lds = h->tune_env_step_lds;
    if (lds < 0) {
        int wgs = h->tune_env_step_waves * 64 / block;                     // workgroups per CU that make the wanted waves
        if (wgs < 1) wgs = 1;
        lds = h->lds_per_cu > 0 ? h->lds_per_cu / wgs / 256 * 256 : 0;
        if (lds > 65536) lds = 65536;                                       // what one workgroup may ask for
        for (int tries = 0; tries < 16 && lds > 256; ++tries) {
            const int nb = occupancy(lds);
            if (nb == 0 || nb == wgs) break;                                // (0: no calculator -- keep the arithmetic figure)
            if (nb < wgs) lds -= 256;                                       // a granule rounded it up past the share
            else if (lds + 256 <= 65536 && occupancy(lds + 256) >= wgs) lds += 256;
            else break;
        }
    }
    h->env_step_lds_resolved = lds;
    h->env_step_wpc = occupancy(lds) * block / 64;
    h->env_step_key[0] = block; h->env_step_key[1] = h->tune_env_step_lds; h->env_step_key[2] = h->tune_env_step_waves;
}

static EnvStepShape env_step_shape(ses_handle *h, int n4)
{
    EnvStepShape sh;
    env_step_resolve(h);
    sh.block = h->tune_env_step_block;
    sh.lds = h->env_step_lds_resolved;
    const long long want = ceil_div((long long)n4, sh.block);
    sh.blocks = want < (1 << 20) ? (int)want : (1 << 20);                   // beyond that the kernel strides
    return sh;
}

}  // namespace ses

extern "C" {

int ses_rollout(ses_handle *h, const float *theta, const float *init, int32_t init_per_offspring, int32_t n_rows,
                int32_t mode, float *fitness, double *ep_return, int32_t *ep_steps)
{
    using namespace ses;
    SES_REQUIRE(h && theta && init && fitness, "ses_rollout: null argument");
    SES_REQUIRE(n_rows >= 1, "ses_rollout: n_rows must be >= 1");
    SES_REQUIRE(mode == SES_MODE_EPISODIC || mode == SES_MODE_FIXED_LENGTH, "ses_rollout: bad mode %d", mode);
    SES_REQUIRE((long long)n_rows * h->cfg.eval_ep_num * 16 < (1ll << 31), "ses_rollout: shard too large");
    SES_REQUIRE(h->cfg.env_id == SES_ENV_CARTPOLE || h->cfg.env_id == SES_ENV_SIMPLE_SPREAD ||
                    h->cfg.env_id == SES_ENV_LUNARLANDER || h->cfg.env_id == SES_ENV_BIPEDALWALKER,
                "ses_rollout: handle has no env");
    SES_HIP_TRY(hipSetDevice(h->cfg.device));
    const size_t episodes = (size_t)n_rows * h->cfg.eval_ep_num;
    double *epr = ep_return;
    if (!epr) {
        int rc = ensure_episode_scratch(h, episodes);
        if (rc != SES_OK) return rc;
        epr = h->ep_return;
    }
    if (h->cfg.env_id == SES_ENV_LUNARLANDER) {
        SES_REQUIRE(mode == SES_MODE_EPISODIC, "ses_rollout: LunarLander has no fixed-length mode");
        const bool epp = h->cfg.gru && !gru_sequential(h) && gru_episode_parallel(h, (long long)episodes);
        if (epp)
            hipLaunchKernelGGL(k_rollout_lander_gru, dim3(ceil_div((long long)episodes, 4)), dim3(256), 0, h->stream,
                               theta, init, init_per_offspring, n_rows, h->cfg.eval_ep_num, h->P, h->cfg.max_step,
                               h->obs_mask, epr, ep_steps, 1);
        else if (h->cfg.gru && !gru_sequential(h) && h->cfg.eval_ep_num >= gru_mfma_min_e(h))
            hipLaunchKernelGGL((k_rollout_gru_mfma<LanderLs, false>), dim3(ceil_div(n_rows, 4)), dim3(256), 0,
                               h->stream, theta, init, init_per_offspring, n_rows, h->cfg.eval_ep_num, h->P,
                               h->cfg.max_step, h->obs_mask, epr, ep_steps);
        else if (h->cfg.gru && !gru_sequential(h) && h->cfg.eval_ep_num <= GL_EB && lander_offspring_per_wave(h, n_rows) == 4)
            hipLaunchKernelGGL((k_rollout_gru_lockstep_multi<LanderLs, 4>), dim3(ceil_div(n_rows, 16)), dim3(256), 0,
                               h->stream, theta, init, init_per_offspring, n_rows, h->cfg.eval_ep_num, h->P,
                               h->cfg.max_step, h->obs_mask, epr, ep_steps);
        else if (h->cfg.gru && !gru_sequential(h) && h->cfg.eval_ep_num <= GL_EB && lander_offspring_per_wave(h, n_rows) == 2)
            hipLaunchKernelGGL((k_rollout_gru_lockstep_multi<LanderLs, 2>), dim3(ceil_div(n_rows, 8)), dim3(256), 0,
                               h->stream, theta, init, init_per_offspring, n_rows, h->cfg.eval_ep_num, h->P,
                               h->cfg.max_step, h->obs_mask, epr, ep_steps);
        else if (h->cfg.gru && !gru_sequential(h))
            hipLaunchKernelGGL((k_rollout_gru_lockstep<LanderLs, false, 1>), dim3(n_rows), dim3(64), 0,
                               h->stream, theta, init, init_per_offspring, n_rows, h->cfg.eval_ep_num, h->P,
                               h->cfg.max_step, h->obs_mask, epr, ep_steps);
        else if (h->cfg.gru)
            hipLaunchKernelGGL(k_rollout_lander_gru, dim3(ceil_div(n_rows, 4)), dim3(256), 0, h->stream, theta, init,
                               init_per_offspring, n_rows, h->cfg.eval_ep_num, h->P, h->cfg.max_step, h->obs_mask, epr,
                               ep_steps, 0);
        else
            launch_box2d_mlp<LanderMlpEnv>(h, theta, init, init_per_offspring, n_rows, epr, ep_steps);
    } else if (h->cfg.env_id == SES_ENV_BIPEDALWALKER) {
        SES_REQUIRE(mode == SES_MODE_EPISODIC, "ses_rollout: BipedalWalker has no fixed-length mode");
        SES_REQUIRE(!h->cfg.gru, "ses_rollout: BipedalWalker has an MLP-policy kernel only (conf/bipedalwalker.yaml: gru False)");
        launch_box2d_mlp<WalkerMlpEnv>(h, theta, init, init_per_offspring, n_rows, epr, ep_steps);
    } else if (h->cfg.env_id == SES_ENV_SIMPLE_SPREAD) {
        SES_REQUIRE(ep_steps == nullptr, "ses_rollout: simple_spread episodes have a fixed length, no ep_steps");
        const int blocks = ceil_div((long long)episodes * 8, 64);
        if (h->cfg.n_agents == 2)
            hipLaunchKernelGGL((k_rollout_spread_mlp<2>), dim3(blocks), dim3(64), 0, h->stream, theta, init,
                               init_per_offspring, n_rows, h->cfg.eval_ep_num, h->P, h->cfg.max_step, epr);
        else
            hipLaunchKernelGGL((k_rollout_spread_mlp<3>), dim3(blocks), dim3(64), 0, h->stream, theta, init,
                               init_per_offspring, n_rows, h->cfg.eval_ep_num, h->P, h->cfg.max_step, epr);
    } else if (h->cfg.physics64) {
        // gym-order float64 dynamics: GRU lockstep or MLP at 4 / 8 lanes per env (parity option, not the bench path)
        const int E = h->cfg.eval_ep_num, T = h->cfg.max_step;
        const bool fixed = mode == SES_MODE_FIXED_LENGTH;
        if (h->cfg.gru) {
            const int blocks = ceil_div(n_rows, 4);
            if (fixed)
                hipLaunchKernelGGL((k_rollout_gru_lockstep<CartPoleLs64, true, 4>), dim3(blocks), dim3(256), 0, h->stream,
                                   theta, init, init_per_offspring, n_rows, E, h->P, T, h->obs_mask, epr, ep_steps);
            else
                hipLaunchKernelGGL((k_rollout_gru_lockstep<CartPoleLs64, false, 4>), dim3(blocks), dim3(256), 0, h->stream,
                                   theta, init, init_per_offspring, n_rows, E, h->P, T, h->obs_mask, epr, ep_steps);
        } else if (pick_lanes_per_env(h, (long long)episodes) >= 8) {
            const int blocks = ceil_div((long long)episodes * 8, 64);
            if (fixed)
                hipLaunchKernelGGL((k_rollout_cartpole_mlp<8, true, 64, true>), dim3(blocks), dim3(64), 0, h->stream, theta,
                                   init, init_per_offspring, n_rows, E, h->P, T, h->obs_mask, epr, ep_steps);
            else
                hipLaunchKernelGGL((k_rollout_cartpole_mlp<8, false, 64, true>), dim3(blocks), dim3(64), 0, h->stream, theta,
                                   init, init_per_offspring, n_rows, E, h->P, T, h->obs_mask, epr, ep_steps);
        } else {
            const int blocks = ceil_div((long long)episodes * 4, 64);
            if (fixed)
                hipLaunchKernelGGL((k_rollout_cartpole_mlp<4, true, 64, true>), dim3(blocks), dim3(64), 0, h->stream, theta,
                                   init, init_per_offspring, n_rows, E, h->P, T, h->obs_mask, epr, ep_steps);
            else
                hipLaunchKernelGGL((k_rollout_cartpole_mlp<4, false, 64, true>), dim3(blocks), dim3(64), 0, h->stream, theta,
                                   init, init_per_offspring, n_rows, E, h->P, T, h->obs_mask, epr, ep_steps);
        }
    } else if (h->cfg.gru && !gru_sequential(h) && gru_episode_parallel(h, (long long)episodes)) {
        const int blocks = ceil_div((long long)episodes, 4);
        if (mode == SES_MODE_FIXED_LENGTH)
            hipLaunchKernelGGL((k_rollout_cartpole_gru<true>), dim3(blocks), dim3(256), 0, h->stream, theta, init,
                               init_per_offspring, n_rows, h->cfg.eval_ep_num, h->P, h->cfg.max_step, h->obs_mask,
                               epr, ep_steps, 1);
        else
            hipLaunchKernelGGL((k_rollout_cartpole_gru<false>), dim3(blocks), dim3(256), 0, h->stream, theta, init,
                               init_per_offspring, n_rows, h->cfg.eval_ep_num, h->P, h->cfg.max_step, h->obs_mask,
                               epr, ep_steps, 1);
    } else if (h->cfg.gru && !gru_sequential(h) && gru_mfma4(h)) {
        // 4x4x1 MFMA blocks (ses_gru_mfma4.h): the policy step costs the same for any eval_ep_num up to 8
        const int blocks = ceil_div(n_rows, 4);
        if (mode == SES_MODE_FIXED_LENGTH)
            hipLaunchKernelGGL((k_rollout_gru_mfma4<CartPoleLs, true>), dim3(blocks), dim3(256), 0, h->stream, theta, init,
                               init_per_offspring, n_rows, h->cfg.eval_ep_num, h->P, h->cfg.max_step, h->obs_mask, epr, ep_steps);
        else
            hipLaunchKernelGGL((k_rollout_gru_mfma4<CartPoleLs, false>), dim3(blocks), dim3(256), 0, h->stream, theta, init,
                               init_per_offspring, n_rows, h->cfg.eval_ep_num, h->P, h->cfg.max_step, h->obs_mask, epr, ep_steps);
    } else if (h->cfg.gru && !gru_sequential(h) && h->cfg.eval_ep_num >= gru_mfma_min_e(h)) {
        const int blocks = ceil_div(n_rows, 4);
        if (mode == SES_MODE_FIXED_LENGTH)
            hipLaunchKernelGGL((k_rollout_gru_mfma<CartPoleLs, true>), dim3(blocks), dim3(256), 0, h->stream, theta,
                               init, init_per_offspring, n_rows, h->cfg.eval_ep_num, h->P, h->cfg.max_step, h->obs_mask,
                               epr, ep_steps);
        else
            hipLaunchKernelGGL((k_rollout_gru_mfma<CartPoleLs, false>), dim3(blocks), dim3(256), 0, h->stream, theta,
                               init, init_per_offspring, n_rows, h->cfg.eval_ep_num, h->P, h->cfg.max_step, h->obs_mask,
                               epr, ep_steps);
    } else if (h->cfg.gru && !gru_sequential(h)) {
        const int blocks = ceil_div(n_rows, 4);
        if (mode == SES_MODE_FIXED_LENGTH)
            hipLaunchKernelGGL((k_rollout_gru_lockstep<CartPoleLs, true, 4>), dim3(blocks), dim3(256), 0, h->stream, theta,
                               init, init_per_offspring, n_rows, h->cfg.eval_ep_num, h->P, h->cfg.max_step, h->obs_mask,
                               epr, ep_steps);
        else
            hipLaunchKernelGGL((k_rollout_gru_lockstep<CartPoleLs, false, 4>), dim3(blocks), dim3(256), 0, h->stream, theta,
                               init, init_per_offspring, n_rows, h->cfg.eval_ep_num, h->P, h->cfg.max_step, h->obs_mask,
                               epr, ep_steps);
    } else if (h->cfg.gru) {
        const int blocks = ceil_div(n_rows, 4);
        if (mode == SES_MODE_FIXED_LENGTH)
            hipLaunchKernelGGL((k_rollout_cartpole_gru<true>), dim3(blocks), dim3(256), 0, h->stream, theta, init,
                               init_per_offspring, n_rows, h->cfg.eval_ep_num, h->P, h->cfg.max_step, h->obs_mask,
                               epr, ep_steps, 0);
        else
            hipLaunchKernelGGL((k_rollout_cartpole_gru<false>), dim3(blocks), dim3(256), 0, h->stream, theta, init,
                               init_per_offspring, n_rows, h->cfg.eval_ep_num, h->P, h->cfg.max_step, h->obs_mask,
                               epr, ep_steps, 0);
    } else {
        launch_cartpole_mlp(h, theta, init, init_per_offspring, n_rows, mode, epr, ep_steps);
    }
    if (h->skip_mean) {
        // (ses_run_generations on one GPU: the counting rank of the tail forms the means itself, k_rank_count_episodes)
        SES_REQUIRE(epr == h->ep_return, "ses_rollout: the fused episode mean works on the handle's own episode scratch");
    } else if (h->fit_gv)
        hipLaunchKernelGGL(k_fitness_mean_granules, dim3(ceil_div(n_rows, 256)), dim3(256), 0, h->stream, epr, n_rows,
                           h->cfg.eval_ep_num, fitness, h->stamp, *h->fit_gv);
    else
        hipLaunchKernelGGL(k_fitness_mean, dim3(ceil_div(n_rows, 256)), dim3(256), 0, h->stream, epr, n_rows,
                           h->cfg.eval_ep_num, fitness, h->stamp);
    SES_HIP_TRY(hipGetLastError());
    return SES_OK;
}

int ses_env_step(ses_handle *h, int32_t n, int32_t mode, float *x, float *xd, float *th, float *thd,
                 const int32_t *action, float *ret, uint32_t *status)
{
    using namespace ses;
    SES_REQUIRE(h && x && xd && th && thd && action && ret && status, "ses_env_step: null argument");
    SES_REQUIRE(n >= 1, "ses_env_step: n must be >= 1");
    SES_REQUIRE(mode == SES_MODE_EPISODIC || mode == SES_MODE_FIXED_LENGTH, "ses_env_step: bad mode %d", mode);
    SES_REQUIRE(h->cfg.env_id == SES_ENV_CARTPOLE, "ses_env_step: handle has no env");
    SES_HIP_TRY(hipSetDevice(h->cfg.device));
    const uintptr_t align = (uintptr_t)x | (uintptr_t)xd | (uintptr_t)th | (uintptr_t)thd | (uintptr_t)action |
                            (uintptr_t)ret | (uintptr_t)status;
    const int n4 = (align & 15u) ? 0 : n / 4;  // unaligned arrays take the scalar path entirely
    const int max_step = h->cfg.max_step;
    if (n4 > 0) {
        // One float4 group per thread, a one-shot grid -- and FEW WAVES IN FLIGHT: single-wave workgroups that each reserve
        // a seventh of the CU's LDS (22.5 KB of 160 on gfx950; derived from the device, env_step_resolve) without touching
        // it, so that a CU holds 7 of them instead of 32 waves.  Thirteen streams from 8192
        // resident waves thrash the memory system's open pages; from 1792 they do not: 2^24 envs, same box, interleaved
        // (tools/envstep_ab.hip): 256 threads / no limit 152.5 us = 0.715 of 8 TB/s; 5 / 6 / 7 / 8 / 10 / 12 / 16 waves per CU
        // 148.0 / 132.1 / 132.1 / 133.9 / 134.4 / 141.1 / 151.1 us -- 0.826 at 6-7, above a plain two-stream copy (0.805).
        // ses_set_tuning "env_step_block" / "env_step_waves_per_cu" / "env_step_lds_bytes" (-1 = derive; 0 with block 256 = the
        // old shape: no reservation).
        const EnvStepShape sh = env_step_shape(h, n4);
        if (mode == SES_MODE_FIXED_LENGTH)
            hipLaunchKernelGGL((k_env_step_cartpole_v4<true>), dim3(sh.blocks), dim3(sh.block), sh.lds, h->stream, n4, max_step,
                               (f32x4 *)x, (f32x4 *)xd, (f32x4 *)th, (f32x4 *)thd, (const i32x4 *)action,
                               (f32x4 *)ret, (u32x4 *)status);
        else
            hipLaunchKernelGGL((k_env_step_cartpole_v4<false>), dim3(sh.blocks), dim3(sh.block), sh.lds, h->stream, n4, max_step,
                               (f32x4 *)x, (f32x4 *)xd, (f32x4 *)th, (f32x4 *)thd, (const i32x4 *)action,
                               (f32x4 *)ret, (u32x4 *)status);
    }
    const int first = n4 * 4;
    if (first < n) {
        const int blocks = ceil_div(n - first, 256);
        if (mode == SES_MODE_FIXED_LENGTH)
            hipLaunchKernelGGL((k_env_step_cartpole_scalar<true>), dim3(blocks), dim3(256), 0, h->stream, first, n,
                               max_step, x, xd, th, thd, action, ret, status);
        else
            hipLaunchKernelGGL((k_env_step_cartpole_scalar<false>), dim3(blocks), dim3(256), 0, h->stream, first, n,
                               max_step, x, xd, th, thd, action, ret, status);
    }
    SES_HIP_TRY(hipGetLastError());
    return SES_OK;
}

int ses_env_step_shape(ses_handle *h, int32_t *block, int32_t *lds_bytes, int32_t *waves_per_cu, int32_t *lds_per_cu)
{
    using namespace ses;
    SES_REQUIRE(h, "ses_env_step_shape: null handle");
    SES_HIP_TRY(hipSetDevice(h->cfg.device));
    env_step_resolve(h);
    if (block) *block = h->tune_env_step_block;
    if (lds_bytes) *lds_bytes = h->env_step_lds_resolved;
    if (waves_per_cu) *waves_per_cu = h->env_step_wpc;
    if (lds_per_cu) *lds_per_cu = h->lds_per_cu;
    return SES_OK;
}

int ses_stream_probe(ses_handle *h, int32_t n, float *x, float *xd, float *th, float *thd, const int32_t *action, float *ret,
                     uint32_t *status)
{
    using namespace ses;
    SES_REQUIRE(h && x && xd && th && thd && action && ret && status, "ses_stream_probe: null argument");
    const uintptr_t align = (uintptr_t)x | (uintptr_t)xd | (uintptr_t)th | (uintptr_t)thd | (uintptr_t)action |
                            (uintptr_t)ret | (uintptr_t)status;
    SES_REQUIRE(n >= 4 && (n & 3) == 0 && (align & 15u) == 0, "ses_stream_probe: n must be a multiple of 4 and the arrays 16-byte aligned");
    SES_HIP_TRY(hipSetDevice(h->cfg.device));
    const int n4 = n / 4;
    const EnvStepShape sh = env_step_shape(h, n4);                                       // the env-step kernel's launch shape
    hipLaunchKernelGGL(k_stream_probe13, dim3(sh.blocks), dim3(sh.block), sh.lds, h->stream, n4, (f32x4 *)x, (f32x4 *)xd, (f32x4 *)th,
                       (f32x4 *)thd, (const i32x4 *)action, (f32x4 *)ret, (u32x4 *)status);
    SES_HIP_TRY(hipGetLastError());
    return SES_OK;
}

int ses_policy_forward(ses_handle *h, const float *theta, const float *obs, float *hidden, int32_t n, float *logits,
                       float *act, int32_t *action)
{
    using namespace ses;
    SES_REQUIRE(h && theta && obs && logits && action, "ses_policy_forward: null argument");
    SES_REQUIRE(n >= 1 && (long long)n * 4 < (1ll << 31), "ses_policy_forward: n out of range");
    SES_HIP_TRY(hipSetDevice(h->cfg.device));
    const int S = h->cfg.num_state, A = h->cfg.num_action;
    if (h->cfg.gru) {
        SES_REQUIRE(hidden, "ses_policy_forward: GRU policy needs the hidden-state array");
        const int gblocks = ceil_div(n, 4);
#define SES_GRU_CASE(S_, A_)                                                                                      \
    if (S == S_ && A == A_) {                                                                                     \
        hipLaunchKernelGGL((k_policy_forward_gru<S_, A_>), dim3(gblocks), dim3(256), 0, h->stream, theta, obs, hidden, n, \
                           h->P, logits, act, action);                                                            \
        SES_HIP_TRY(hipGetLastError());                                                                           \
        return SES_OK;                                                                                            \
    }
        SES_GRU_CASE(4, 2)
        SES_GRU_CASE(8, 4)
#undef SES_GRU_CASE
        return set_error(SES_ERR_UNSUPPORTED, "ses_policy_forward: no GRU kernel instance for num_state=%d num_action=%d", S, A);
    }
    const int blocks = ceil_div((long long)n * 4, 64);
#define SES_FWD_CASE(S_, A_)                                                                                      \
    if (S == S_ && A == A_) {                                                                                     \
        hipLaunchKernelGGL((k_policy_forward_mlp<S_, A_>), dim3(blocks), dim3(64), 0, h->stream, theta, obs, n, h->P, \
                           logits, act, action);                                                                  \
        SES_HIP_TRY(hipGetLastError());                                                                           \
        return SES_OK;                                                                                            \
    }
    SES_FWD_CASE(4, 2)
    SES_FWD_CASE(8, 4)
    SES_FWD_CASE(12, 5)
    SES_FWD_CASE(18, 5)
    SES_FWD_CASE(24, 4)
#undef SES_FWD_CASE
    return set_error(SES_ERR_UNSUPPORTED, "ses_policy_forward: no kernel instance for num_state=%d num_action=%d", S, A);
}

}  // extern "C"

#ifdef SES_PHASE_TIMERS
// development build only (tools/walker_phases.py): the phase totals of ses_lander.h's phase_mark, read and optionally cleared
extern "C" int ses_debug_phase_totals(unsigned long long *out24, int reset)
{
    if (hipMemcpyFromSymbol(out24, HIP_SYMBOL(ses::phase_total), ses::PHASE_SLOTS * sizeof(unsigned long long)) != hipSuccess) return -1;
    if (reset) {
        unsigned long long z[ses::PHASE_SLOTS] = {};
        if (hipMemcpyToSymbol(HIP_SYMBOL(ses::phase_total), z, sizeof z) != hipSuccess) return -1;
    }
    return 0;
}
#endif
