// ses_envs.hip -- step-wise device entry for EVERY env of the library: one lane = one env, state in an opaque
// caller-owned blob (ses_env_state_bytes per env).  This is `env.reset()` / `env.step(action)` of the reference's
// wrappers (envs/gym_wrapper.py:23-45, envs/pettingzoo_wrapper.py:22-58) for n independent envs at once: the kernels
// call the SAME device functions the fused rollouts call (cartpole_step_general, ll_step, bw_step, spread_step), so an
// env can be stepped, inspected and compared with the oracle's env objects one transition at a time, and the
// reference's playback loop (test.py:53-63) runs against the wrappers for every supported env.
//
// Blob layout (opaque to the caller, fixed per handle): CartPole {x, xd, th, thd}; simple_spread SpreadState<NA> + the
// cycle counter; LunarLander / BipedalWalker the env struct of the Box2D-style world followed by the episode's terrain
// heights (the fused rollouts keep those in LDS; here they live in the blob and the env reads them through the same
// pointer).  Truncation at env.max_step is the wrapper's job (gym_wrapper.py:37-39), as in the reference.
#include "ses_cartpole.h"
#include "ses_internal.h"
#include "ses_lander.h"
#include "ses_policy.h"
#include "ses_spread.h"
#include "ses_walker.h"

namespace ses {

struct CartPoleBlob {
    CartPoleState st;
};

template <int NA>
struct SpreadBlob {
    SpreadState<NA> st;
    int32_t cycle;
};

struct LanderBlob {
    b2l::LanderEnv env;
    float ty[LL_TERRAIN_ROW];
};

struct WalkerBlob {
    b2l::WalkerEnv env;
    float ty[BW_TERRAIN_ROW];
};

constexpr int SPREAD_MAX_CYCLES = 25;        // pettingzoo mpe default max_cycles: every agent is done after 25 cycles

__device__ __forceinline__ void store_obs_masked(float *__restrict__ dst, const float *obs, int S, uint32_t obs_mask)
{
    for (int k = 0; k < S; ++k) dst[k] = ((obs_mask >> k) & 1u) ? 0.0f : obs[k];
}

// ---- CartPole --------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void k_envs_reset_cartpole(const float *__restrict__ init, int n, CartPoleBlob *__restrict__ state,
                                                            float *__restrict__ obs, uint32_t obs_mask)
{
    const int i = blockIdx.x * 64 + threadIdx.x;
    if (i >= n) return;
    const float *u = init + (size_t)i * 4;
    const CartPoleState st{u[0], u[1], u[2], u[3]};
    state[i].st = st;
    const float o[4] = {st.x, st.xd, st.th, st.thd};
    store_obs_masked(obs + (size_t)i * 4, o, 4, obs_mask);
}

__global__ __launch_bounds__(64) void k_envs_step_cartpole(CartPoleBlob *__restrict__ state, const int32_t *__restrict__ action, int n,
                                                           float *__restrict__ obs, float *__restrict__ reward,
                                                           int32_t *__restrict__ done, uint32_t obs_mask)
{
    const int i = blockIdx.x * 64 + threadIdx.x;
    if (i >= n) return;
    CartPoleState st = state[i].st;
    const bool term = cartpole_step_general(st, action[i]);
    state[i].st = st;
    const float o[4] = {st.x, st.xd, st.th, st.thd};
    store_obs_masked(obs + (size_t)i * 4, o, 4, obs_mask);
    reward[i] = 1.0f;
    done[i] = term ? 1 : 0;
}

// ---- simple_spread -----------------------------------------------------------------------------------------------
template <int NA>
__device__ __forceinline__ void spread_store_obs(const SpreadState<NA> &st, float *__restrict__ dst)
{
#pragma unroll
    for (int a = 0; a < NA; ++a) {
        float o[6 * NA];
        spread_obs<NA>(st, a, o);
#pragma unroll
        for (int k = 0; k < 6 * NA; ++k) dst[a * 6 * NA + k] = o[k];
    }
}

template <int NA>
__global__ __launch_bounds__(64) void k_envs_reset_spread(const float *__restrict__ init, int n, SpreadBlob<NA> *__restrict__ state,
                                                          float *__restrict__ obs)
{
    const int i = blockIdx.x * 64 + threadIdx.x;
    if (i >= n) return;
    const float *s0 = init + (size_t)i * (4 * NA);
    SpreadState<NA> st;
#pragma unroll
    for (int a = 0; a < NA; ++a) {
        st.ax[a] = s0[2 * a]; st.ay[a] = s0[2 * a + 1];
        st.vx[a] = 0.0f; st.vy[a] = 0.0f;
        st.lx[a] = s0[2 * NA + 2 * a]; st.ly[a] = s0[2 * NA + 2 * a + 1];
    }
    state[i].st = st;
    state[i].cycle = 0;
    spread_store_obs<NA>(st, obs + (size_t)i * NA * 6 * NA);
}

template <int NA>
__global__ __launch_bounds__(64) void k_envs_step_spread(SpreadBlob<NA> *__restrict__ state, const int32_t *__restrict__ action, int n,
                                                         float *__restrict__ obs, float *__restrict__ reward,
                                                         int32_t *__restrict__ done)
{
    const int i = blockIdx.x * 64 + threadIdx.x;
    if (i >= n) return;
    SpreadState<NA> st = state[i].st;
    int act[NA];
#pragma unroll
    for (int a = 0; a < NA; ++a) act[a] = action[(size_t)i * NA + a];
    const float r = spread_step<NA>(st, act);
    const int cycle = state[i].cycle + 1;
    state[i].st = st;
    state[i].cycle = cycle;
    spread_store_obs<NA>(st, obs + (size_t)i * NA * 6 * NA);
    reward[i] = r;
    done[i] = cycle >= SPREAD_MAX_CYCLES ? 1 : 0;
}

// ---- the Box2D-style envs ------------------------------------------------------------------------------------------
// One lane per env: a wave steps 64 different worlds at once (ll_step / bw_step contain no wave-level operation).
__global__ __launch_bounds__(64, 2) void k_envs_reset_lander(const float *__restrict__ init, int n, LanderBlob *__restrict__ state,
                                                             float *__restrict__ obs, uint32_t obs_mask)
{
    const int i = blockIdx.x * 64 + threadIdx.x;
    if (i >= n) return;
    LanderBlob &b = state[i];
    LanderState s;
    ll_reset(s, init + (size_t)i * 16, b.ty);                  // the terrain row is the blob's own (ll_reset ends with the no-op step)
    b.env = s.env;
    float o[8];
    ll_obs(s, o);
    store_obs_masked(obs + (size_t)i * 8, o, 8, obs_mask);
}

__global__ __launch_bounds__(64, 2) void k_envs_step_lander(LanderBlob *__restrict__ state, const float *__restrict__ action, int A, int n,
                                                            float *__restrict__ obs, float *__restrict__ reward,
                                                            int32_t *__restrict__ done, uint32_t obs_mask)
{
    const int i = blockIdx.x * 64 + threadIdx.x;
    if (i >= n) return;
    LanderBlob &b = state[i];
    LanderState s;
    s.env = b.env;
    s.ty = b.ty;
    bool d;
    const float r = ll_step(s, action[(size_t)i * A], action[(size_t)i * A + 1], d);   // the env uses outputs 0 and 1
    b.env = s.env;
    float o[8];
    ll_obs(s, o);
    store_obs_masked(obs + (size_t)i * 8, o, 8, obs_mask);
    reward[i] = r;
    done[i] = d ? 1 : 0;
}

__global__ __launch_bounds__(64, 1) void k_envs_reset_walker(const float *__restrict__ init, int n, WalkerBlob *__restrict__ state,
                                                             float *__restrict__ obs)
{
    const int i = blockIdx.x * 64 + threadIdx.x;
    if (i >= n) return;
    WalkerBlob &b = state[i];
    WalkerState s;
    bw_reset(s, init + (size_t)i * 4, b.ty);
    b.env = s.env;
    float o[24];
    bw_obs(s, o);
    store_obs_masked(obs + (size_t)i * 24, o, 24, 0u);
}

__global__ __launch_bounds__(64, 1) void k_envs_step_walker(WalkerBlob *__restrict__ state, const float *__restrict__ action, int n,
                                                            float *__restrict__ obs, float *__restrict__ reward,
                                                            int32_t *__restrict__ done)
{
    const int i = blockIdx.x * 64 + threadIdx.x;
    if (i >= n) return;
    WalkerBlob &b = state[i];
    WalkerState s;
    s.env = b.env;
    s.ty = b.ty;
    const float *a = action + (size_t)i * 4;
    bool d;
    const float r = bw_step(s, a[0], a[1], a[2], a[3], d);
    b.env = s.env;
    float o[24];
    bw_obs(s, o);
    store_obs_masked(obs + (size_t)i * 24, o, 24, 0u);
    reward[i] = r;
    done[i] = d ? 1 : 0;
}

static int env_state_bytes(const ses_handle *h)
{
    switch (h->cfg.env_id) {
        case SES_ENV_CARTPOLE: return (int)sizeof(CartPoleBlob);
        case SES_ENV_SIMPLE_SPREAD: return h->cfg.n_agents == 2 ? (int)sizeof(SpreadBlob<2>) : (int)sizeof(SpreadBlob<3>);
        case SES_ENV_LUNARLANDER: return (int)sizeof(LanderBlob);
        case SES_ENV_BIPEDALWALKER: return (int)sizeof(WalkerBlob);
        default: return 0;
    }
}

}  // namespace ses

extern "C" {

int ses_env_state_bytes(ses_handle *h)
{
    using namespace ses;
    SES_REQUIRE(h, "ses_env_state_bytes: null handle");
    const int b = env_state_bytes(h);
    SES_REQUIRE(b > 0, "ses_env_state_bytes: handle has no env");
    return b;
}

int ses_env_obs_width(ses_handle *h)
{
    using namespace ses;
    SES_REQUIRE(h, "ses_env_obs_width: null handle");
    switch (h->cfg.env_id) {
        case SES_ENV_CARTPOLE: return 4;
        case SES_ENV_SIMPLE_SPREAD: return h->cfg.n_agents * 6 * h->cfg.n_agents;
        case SES_ENV_LUNARLANDER: return 8;
        case SES_ENV_BIPEDALWALKER: return 24;
        default: return set_error(SES_ERR_INVALID_ARG, "ses_env_obs_width: handle has no env");
    }
}

int ses_env_reset(ses_handle *h, const float *init, int32_t n, void *state, float *obs)
{
    using namespace ses;
    SES_REQUIRE(h && init && state && obs, "ses_env_reset: null argument");
    SES_REQUIRE(n >= 1, "ses_env_reset: n must be >= 1");
    SES_REQUIRE(env_state_bytes(h) > 0, "ses_env_reset: handle has no env");
    SES_REQUIRE(!h->cfg.physics64, "ses_env_reset: the step-wise CartPole is the float32 one (physics64 exists in the fused rollouts only)");
    SES_HIP_TRY(hipSetDevice(h->cfg.device));
    const dim3 grid(ceil_div(n, 64)), block(64);
    switch (h->cfg.env_id) {
        case SES_ENV_CARTPOLE:
            hipLaunchKernelGGL(k_envs_reset_cartpole, grid, block, 0, h->stream, init, n, (CartPoleBlob *)state, obs, h->obs_mask);
            break;
        case SES_ENV_SIMPLE_SPREAD:
            if (h->cfg.n_agents == 2)
                hipLaunchKernelGGL(k_envs_reset_spread<2>, grid, block, 0, h->stream, init, n, (SpreadBlob<2> *)state, obs);
            else
                hipLaunchKernelGGL(k_envs_reset_spread<3>, grid, block, 0, h->stream, init, n, (SpreadBlob<3> *)state, obs);
            break;
        case SES_ENV_LUNARLANDER:
            hipLaunchKernelGGL(k_envs_reset_lander, grid, block, 0, h->stream, init, n, (LanderBlob *)state, obs, h->obs_mask);
            break;
        default:
            hipLaunchKernelGGL(k_envs_reset_walker, grid, block, 0, h->stream, init, n, (WalkerBlob *)state, obs);
            break;
    }
    SES_HIP_TRY(hipGetLastError());
    return SES_OK;
}

int ses_env_step_generic(ses_handle *h, void *state, const void *action, int32_t n, float *obs, float *reward, int32_t *done)
{
    using namespace ses;
    SES_REQUIRE(h && state && action && obs && reward && done, "ses_env_step_generic: null argument");
    SES_REQUIRE(n >= 1, "ses_env_step_generic: n must be >= 1");
    SES_REQUIRE(env_state_bytes(h) > 0, "ses_env_step_generic: handle has no env");
    SES_HIP_TRY(hipSetDevice(h->cfg.device));
    const dim3 grid(ceil_div(n, 64)), block(64);
    switch (h->cfg.env_id) {
        case SES_ENV_CARTPOLE:
            hipLaunchKernelGGL(k_envs_step_cartpole, grid, block, 0, h->stream, (CartPoleBlob *)state, (const int32_t *)action, n, obs,
                               reward, done, h->obs_mask);
            break;
        case SES_ENV_SIMPLE_SPREAD:
            if (h->cfg.n_agents == 2)
                hipLaunchKernelGGL(k_envs_step_spread<2>, grid, block, 0, h->stream, (SpreadBlob<2> *)state, (const int32_t *)action, n,
                                   obs, reward, done);
            else
                hipLaunchKernelGGL(k_envs_step_spread<3>, grid, block, 0, h->stream, (SpreadBlob<3> *)state, (const int32_t *)action, n,
                                   obs, reward, done);
            break;
        case SES_ENV_LUNARLANDER:
            hipLaunchKernelGGL(k_envs_step_lander, grid, block, 0, h->stream, (LanderBlob *)state, (const float *)action,
                               h->cfg.num_action, n, obs, reward, done, h->obs_mask);
            break;
        default:
            hipLaunchKernelGGL(k_envs_step_walker, grid, block, 0, h->stream, (WalkerBlob *)state, (const float *)action, n, obs,
                               reward, done);
            break;
    }
    SES_HIP_TRY(hipGetLastError());
    return SES_OK;
}

}  // extern "C"
