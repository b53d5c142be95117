// ses_core.hip -- handle lifecycle and error reporting of libses_hip.so (see include/ses.h)
#include <cstring>
#include <string>

#include "ses_internal.h"

namespace ses {

static thread_local char g_err[512] = "";

int set_error(int code, const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
    return code;
}

int ensure_episode_scratch(ses_handle *h, size_t episodes)
{
    if (episodes <= h->ep_cap) return SES_OK;
    if (h->ep_return || h->ep_steps) SES_HIP_TRY(hipStreamSynchronize(h->stream));   // kernels may still use the old buffers
    if (h->ep_return) SES_HIP_TRY(hipFree(h->ep_return));
    if (h->ep_steps) SES_HIP_TRY(hipFree(h->ep_steps));
    h->ep_return = nullptr;
    h->ep_steps = nullptr;
    h->ep_cap = 0;
    SES_HIP_TRY(hipMalloc(&h->ep_return, episodes * sizeof(double)));
    SES_HIP_TRY(hipMalloc(&h->ep_steps, episodes * sizeof(int32_t)));
    h->ep_cap = episodes;
    return SES_OK;
}

int ensure_reduce_scratch(ses_handle *h, size_t bytes)
{
    if (bytes <= h->red_cap) return SES_OK;
    // grow generously: the buffer is re-used by every generation and a free/alloc pair would stall the stream
    size_t want = bytes < (1u << 20) ? (1u << 20) : bytes * 2;
    if (h->red_scratch) {
        SES_HIP_TRY(hipStreamSynchronize(h->stream));
        SES_HIP_TRY(hipFree(h->red_scratch));
    }
    h->red_scratch = nullptr;
    h->red_cap = 0;
    h->rank_zeroed = nullptr;                       // whatever was known about the old buffer's contents is gone
    h->counter_armed = nullptr;
    SES_HIP_TRY(hipMalloc(&h->red_scratch, want));
    h->red_cap = want;
    return SES_OK;
}

}  // namespace ses

extern "C" {

const char *ses_last_error(void) { return ses::g_err; }

const char *ses_version(void) { return "ses-hip 0.1 (gfx950)"; }

int ses_param_count(int32_t S, int32_t A, int32_t gru)
{
    int p = SES_HIDDEN * S + SES_HIDDEN + A * SES_HIDDEN + A;
    if (gru) p += 2 * (3 * SES_HIDDEN * SES_HIDDEN) + 2 * (3 * SES_HIDDEN);
    return p;
}

int ses_device_count(void)
{
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) return ses::set_error(SES_ERR_NO_DEVICE, "hipGetDeviceCount: %s", hipGetErrorString(e));
    return n;
}

int ses_create(const ses_config *cfg, void *stream, ses_handle **out)
{
    SES_REQUIRE(cfg && out, "ses_create: null argument");
    SES_REQUIRE(cfg->env_id == SES_ENV_CARTPOLE || cfg->env_id == SES_ENV_NONE || cfg->env_id == SES_ENV_SIMPLE_SPREAD ||
                    cfg->env_id == SES_ENV_LUNARLANDER || cfg->env_id == SES_ENV_BIPEDALWALKER,
                "ses_create: unknown env_id %d", cfg->env_id);
    SES_REQUIRE(cfg->num_state >= 1 && cfg->num_state <= 32, "ses_create: num_state %d out of range", cfg->num_state);
    SES_REQUIRE(cfg->num_action >= 1 && cfg->num_action <= 8, "ses_create: num_action %d out of range", cfg->num_action);
    SES_REQUIRE(cfg->eval_ep_num >= 1, "ses_create: eval_ep_num must be >= 1");
    SES_REQUIRE(cfg->max_step >= 1 && cfg->max_step < (1 << 30), "ses_create: max_step must be in [1, 2^30)");
    SES_REQUIRE(cfg->lanes_per_env == 0 || cfg->lanes_per_env == 1 || cfg->lanes_per_env == 2 ||
                    cfg->lanes_per_env == 4 || cfg->lanes_per_env == 8 || cfg->lanes_per_env == 16 || cfg->lanes_per_env == 32,
                "ses_create: lanes_per_env must be 0, 1, 2, 4, 8, 16 or 32");
    if (cfg->env_id == SES_ENV_CARTPOLE)
        SES_REQUIRE(cfg->num_state == 4 && cfg->num_action == 2 && cfg->discrete_action,
                    "ses_create: CartPole needs num_state=4 num_action=2 discrete_action=1");
    if (cfg->env_id == SES_ENV_LUNARLANDER)
        SES_REQUIRE(cfg->num_state == 8 && cfg->num_action == 4 && !cfg->discrete_action,
                    "ses_create: LunarLanderContinuous needs num_state=8 num_action=4 discrete_action=0");
    if (cfg->env_id == SES_ENV_BIPEDALWALKER)
        SES_REQUIRE(cfg->num_state == 24 && cfg->num_action == 4 && !cfg->discrete_action && !cfg->gru && !cfg->pomdp,
                    "ses_create: BipedalWalker needs num_state=24 num_action=4 discrete_action=0 gru=0 pomdp=0");
    if (cfg->env_id == SES_ENV_SIMPLE_SPREAD)
        SES_REQUIRE((cfg->n_agents == 2 || cfg->n_agents == 3) && cfg->num_state == 6 * cfg->n_agents &&
                        cfg->num_action == 5 && cfg->discrete_action && !cfg->gru,
                    "ses_create: simple_spread needs n_agents in {2,3}, num_state=6*n_agents, num_action=5, "
                    "discrete_action=1, gru=0");
    SES_REQUIRE(cfg->physics64 == 0 || (cfg->physics64 == 1 && cfg->env_id == SES_ENV_CARTPOLE),
                "ses_create: physics64 is a CartPole rollout option");
    int ndev = ses_device_count();
    if (ndev <= 0) return ses::set_error(SES_ERR_NO_DEVICE, "ses_create: no HIP device visible");
    SES_REQUIRE(cfg->device >= 0 && cfg->device < ndev, "ses_create: device %d not in [0,%d)", cfg->device, ndev);
    SES_HIP_TRY(hipSetDevice(cfg->device));
    ses_handle *h = new ses_handle();
    std::memset(h, 0, sizeof *h);
    h->cfg = *cfg;
    h->stream = (hipStream_t)stream;
    h->P = ses_param_count(cfg->num_state, cfg->num_action, cfg->gru);
    h->obs_mask = 0u;
    if (cfg->pomdp && cfg->env_id == SES_ENV_CARTPOLE) h->obs_mask = 0xAu;      // obs[1], obs[3]  (gym_wrapper.py:73-77)
    if (cfg->pomdp && cfg->env_id == SES_ENV_LUNARLANDER) h->obs_mask = 0x2Cu;  // obs[2,3,5]      (gym_wrapper.py:61-66)
    h->tune_rollout_block = 64;
    h->tune_gru_mfma_min_e = 12;
    h->tune_gru_mfma4_min_e = 7;              // measured (profiles/r06_time_gru.txt): 3.05 ms for any E <= 8 against 3.27 / 3.52 ms of the VALU lockstep kernel at 7 / 8
    h->tune_gru_ep_parallel_max = 4096;
    h->tune_gru_sequential = 0;
    h->tune_rollout_mix = 1;
    h->tune_rollout_waves8 = 1024;
    h->tune_rollout_packed = -1;
    h->tune_rollout_mix_8_16 = 1;
    h->tune_lander_per_wave = 0;
    h->tune_box2d_lpe = 0;
    h->tune_box2d_epw = 0;
    h->tune_env_step_block = 64;
    h->tune_env_step_lds = -1;                // derived from the device: lds_per_cu / tune_env_step_waves (ses_rollout.hip: env_step_shape)
    h->tune_env_step_waves = 7;
    h->env_step_key[0] = -2;
    {
        int lds = 0;
        const hipError_t e = hipDeviceGetAttribute(&lds, hipDeviceAttributeMaxSharedMemoryPerMultiprocessor, cfg->device);
        if (e != hipSuccess) {
            delete h;                         // nothing else is owned yet
            return ses::set_error(SES_ERR_HIP, "ses_create: hipDeviceGetAttribute: %s", hipGetErrorString(e));
        }
        h->lds_per_cu = lds;
    }
    h->tune_openai_sharded_tail = 1;
    h->tune_openai_sharded_min_rows = 8192;
    h->tune_openai_granules = 1;
    h->tune_comm_granules_enabled = 1;
    h->tune_fused_fitness = 1;
    h->tune_fused_mean = 1;
    h->tune_fused_elite = 1;
    h->tune_fused_apply_perturb = 1;
    h->tune_comm_granules = 0;                // measured: 12.4 us against 6.3 for the kernel with sequence words (4096 floats, two ranks)
    h->tune_es_final_max_chunks = 0;          // measured: the wave-per-parameter update launch beats the in-kernel finisher
    *out = h;
    return SES_OK;
}

int ses_set_tuning(ses_handle *h, const char *name, int32_t value)
{
    SES_REQUIRE(h && name, "ses_set_tuning: null argument");
    struct Knob {
        const char *name;
        int ses_handle::*field;
        int lo, hi;
    };
    static const Knob knobs[] = {{"rollout_block", &ses_handle::tune_rollout_block, 64, 256},
                                 {"gru_mfma_min_e", &ses_handle::tune_gru_mfma_min_e, 1, 1 << 30},
                                 {"gru_mfma4_min_e", &ses_handle::tune_gru_mfma4_min_e, 0, 8},
                                 {"gru_ep_parallel_max", &ses_handle::tune_gru_ep_parallel_max, 0, 1 << 30},
                                 {"gru_sequential", &ses_handle::tune_gru_sequential, 0, 1},
                                 {"rollout_mix", &ses_handle::tune_rollout_mix, 0, 1},
                                 {"rollout_waves8", &ses_handle::tune_rollout_waves8, 1, 1 << 20},
                                 {"rollout_mix_light", &ses_handle::tune_rollout_mix_light, 0, 16},
                                 {"rollout_lpe32_max_envs", &ses_handle::tune_rollout_lpe32_max, 0, 1 << 30},
                                 {"rollout_packed", &ses_handle::tune_rollout_packed, -1, 1},
                                 {"rollout_mix_8_16", &ses_handle::tune_rollout_mix_8_16, 0, 1},
                                 {"lander_offspring_per_wave", &ses_handle::tune_lander_per_wave, 0, 4},
                                 {"box2d_lanes_per_env", &ses_handle::tune_box2d_lpe, 0, 64},
                                 {"box2d_envs_per_wave", &ses_handle::tune_box2d_epw, 0, 64},
                                 {"env_step_block", &ses_handle::tune_env_step_block, 64, 256},
                                 {"env_step_lds_bytes", &ses_handle::tune_env_step_lds, -1, 65536},
                                 {"env_step_waves_per_cu", &ses_handle::tune_env_step_waves, 1, 32},
                                 {"es_final_max_chunks", &ses_handle::tune_es_final_max_chunks, 0, 1 << 20},
                                 {"comm_force_rccl", &ses_handle::tune_comm_force_rccl, 0, 1},
                                 {"comm_p2p_timeout_ms", &ses_handle::tune_comm_p2p_timeout_ms, 0, 1 << 30},
                                 {"comm_p2p_keep_going", &ses_handle::tune_comm_p2p_keep_going, 0, 1},
                                 {"openai_sharded_tail", &ses_handle::tune_openai_sharded_tail, 0, 1},
                                 {"openai_sharded_min_rows", &ses_handle::tune_openai_sharded_min_rows, 0, 1 << 30},
                                 {"openai_granule_exchange", &ses_handle::tune_openai_granules, 0, 1},
                                 {"comm_granule_allgather", &ses_handle::tune_comm_granules, 0, 1},
                                 {"comm_granules_enabled", &ses_handle::tune_comm_granules_enabled, 0, 1},
                                 {"fused_fitness_exchange", &ses_handle::tune_fused_fitness, 0, 1},
                                 {"fused_episode_mean", &ses_handle::tune_fused_mean, 0, 1},
                                 {"fused_elite_tail", &ses_handle::tune_fused_elite, 0, 1},
                                 {"fused_apply_perturb", &ses_handle::tune_fused_apply_perturb, 0, 1}};
    for (const Knob &k : knobs) {
        if (std::strcmp(k.name, name) == 0) {
            SES_REQUIRE(value >= k.lo && value <= k.hi, "ses_set_tuning: %s = %d outside [%d, %d]", name, value, k.lo, k.hi);
            SES_REQUIRE(k.field != &ses_handle::tune_rollout_block || value == 64 || value == 256,
                        "ses_set_tuning: rollout_block must be 64 or 256");
            SES_REQUIRE(k.field != &ses_handle::tune_env_step_block || value == 64 || value == 128 || value == 256,
                        "ses_set_tuning: env_step_block must be 64, 128 or 256");
            SES_REQUIRE(k.field != &ses_handle::tune_lander_per_wave || value != 3,
                        "ses_set_tuning: lander_offspring_per_wave must be 0, 1, 2 or 4");
            SES_REQUIRE(k.field != &ses_handle::tune_box2d_lpe || (value & (value - 1)) == 0,
                        "ses_set_tuning: box2d_lanes_per_env must be 0 or a power of two up to 64");
            SES_REQUIRE(k.field != &ses_handle::tune_rollout_mix_light || value == 0 || value == 8 || value == 16,
                        "ses_set_tuning: rollout_mix_light must be 0, 8 or 16");
            h->*(k.field) = value;
            if (k.field == &ses_handle::tune_comm_p2p_timeout_ms) ses::comm_p2p_set_timeout(h);   // also for a live mailbox
            return SES_OK;
        }
    }
    return ses::set_error(SES_ERR_INVALID_ARG, "ses_set_tuning: unknown knob '%s'", name);
}

int ses_set_stamp(ses_handle *h, uint64_t *dst)
{
    SES_REQUIRE(h, "ses_set_stamp: null handle");
    h->stamp = (unsigned long long *)dst;
    return SES_OK;
}

int ses_destroy(ses_handle *h)
{
    if (!h) return SES_OK;
    (void)hipSetDevice(h->cfg.device);
    (void)ses::comm_release(h);
    if (h->ep_return) (void)hipFree(h->ep_return);
    if (h->ep_steps) (void)hipFree(h->ep_steps);
    if (h->red_scratch) (void)hipFree(h->red_scratch);
    if (h->gen_init) (void)hipFree(h->gen_init);
    delete h;
    return SES_OK;
}

int ses_sync(ses_handle *h)
{
    SES_REQUIRE(h, "ses_sync: null handle");
    SES_HIP_TRY(hipStreamSynchronize(h->stream));
    return SES_OK;
}

}  // extern "C"
