// ses_internal.h -- handle layout and error plumbing shared by the translation units of libses_hip.so
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>

#include "../../include/ses.h"

namespace ses { struct P2pGranuleView; }
struct ses_handle {
    ses_config cfg;
    hipStream_t stream;
    int P;               // parameters per offspring
    uint32_t obs_mask;   // bit k set: observation component k is zeroed (POMDP wrappers)
    // scratch owned by the handle (grown on demand, never shrunk)
    double *ep_return;   // [rows * E]
    int32_t *ep_steps;   // [rows * E]
    size_t ep_cap;       // capacity in episodes
    void *red_scratch;   // rank keys (u64[n]) followed by es_update partial sums
    size_t red_cap;      // bytes
    float *gen_init;     // ses_run_generations: the env resets of a chunk of generations, drawn in one launch
    size_t gen_init_cap; // floats
    // multi-GPU (ses_comm.hip): RCCL communicator of this rank, null until ses_comm_init
    void *comm;
    int comm_rank, comm_world;
    // peer-store transport of the same exchange (ses_comm_p2p_*): this rank's mailbox and the peers' mapped ones
    struct ses_p2p *p2p;
    // kernel-selection thresholds (ses_set_tuning; the defaults are the measured crossovers, ses_rollout.hip)
    int tune_rollout_block;        // workgroup size of the pure-LPE CartPole MLP rollout: 64 or 256
    int tune_gru_mfma_min_e;       // eval_ep_num from which GRU rollouts run on the MFMA kernel
    int tune_gru_mfma4_min_e;      // eval_ep_num (... 8) from which the CartPole GRU rollout takes the 4x4x1 MFMA step; 0 = never
    int tune_gru_ep_parallel_max;  // (offspring x episode) count up to which GRU rollouts use one wave per episode
    int tune_gru_sequential;       // 1: episode-after-episode GRU kernels only
    int tune_rollout_mix;          // 0: no mixed LPE-8 / LPE-4 split for mid-sized CartPole MLP populations
    int tune_rollout_waves8;       // light waves of the mixed split
    int tune_rollout_mix_light;    // lanes per env of the light waves: 0 = choose, 8, 16
    int tune_rollout_lpe32_max;    // CartPole MLP populations of up to this many envs run at 32 lanes per env (0: never)
    int tune_rollout_mix_8_16;     // 1: the (8 lanes per env on every SIMD + the rest at 16) split is a candidate (round 6)
    int tune_rollout_packed;       // the packed step of lone waves (ses_policy_pk.h): -1 = when every wave has a SIMD to itself, 0 / 1
    // ses_set_stamp: where the next stamped launch of this handle writes the GPU real-time counter (or null)
    unsigned long long *stamp;
    // ses_openai_generation: the rank vector in red_scratch that is known to be zero (left so by the update kernel)
    int32_t *rank_zeroed;
    int rank_zeroed_n;
    unsigned int *counter_armed;   // the last-block ticket counter in red_scratch that is known to be zero
    int tune_comm_force_rccl;      // 1: ses_allgather_fitness ignores an attached peer-store transport (A/B measurements)
    int tune_es_final_max_chunks;  // ses_openai_generation: up to this many 1024-row chunks the gradient kernel applies Adam itself
    int tune_box2d_lpe;            // lanes per env of the Box2D MLP rollout: 0 = by population size, 1 / 2 / 4 / ... / 64
    int tune_env_step_block;       // threads per workgroup of the standalone env-step kernel (64)
    int tune_env_step_lds;         // bytes of LDS each of its workgroups reserves without touching them: limits the waves in flight;
                                   // -1 (default): derived from the device's LDS per CU and tune_env_step_waves
    int tune_env_step_waves;       // waves per CU the derived reservation keeps in flight (7: what the memory system wants)
    // Transient, set by ses_run_generations around the calls of one generation (null otherwise): the granule view of a fitness
    // exchange that the episode-mean kernel feeds (every rank's mailbox gets this rank's fitness values as granules) and the
    // shard form of the tail consumes (k_rank_sort_search polls the tiles it sorts); fit_own: this rank's own fitness values.
    const ses::P2pGranuleView *fit_gv;
    const float *fit_own;
    int fit_per_rank;              // rows per rank slot of that exchange
    // Transient, set by ses_run_generations on ONE GPU: skip_mean -- ses_rollout leaves the episode returns in ep_return and does
    // not launch the episode-mean kernel; mean_src -- the counting rank of ses_openai_generation forms the means itself from that
    // array (k_rank_count_episodes) and writes fitness[]: one launch less per generation.
    int skip_mean;
    const double *mean_src;
    unsigned long long *mean_stamp;   // where that kernel writes the end-of-rollout time stamp (ses_set_stamp's slot of the rollout)
    int tune_fused_elite;          // 1 (default): ses_run_generations on one GPU runs the elite strategies' tail of populations up to 512
                                   // rows (the kernel serves 1024; ONE workgroup counts -- n compares per row -- so the loop stops using it at 512, twice the reference's largest config) as [mean + rank + best + selection] and, simple_evolution, [elite rows + mean]: two launches for seven
    int tune_fused_apply_perturb;  // 1 (default): the replicated openai_es tail of policies up to 1024 parameters applies the update inside the
                                   // launch that writes the next population (k_es_apply_perturb): one launch less per generation
    int tune_fused_mean;           // 1 (default): ses_run_generations uses the above for openai_es up to 8192 rows on one GPU
    int lds_per_cu;                // hipDeviceAttributeMaxSharedMemoryPerMultiprocessor of the handle's device
    int env_step_key[3];           // (block, lds knob, waves knob) the two values below were resolved for
    int env_step_lds_resolved;     // the reservation actually launched with
    int env_step_wpc;              // waves per CU the occupancy calculator gives that shape
    int tune_box2d_epw;            // different envs per wave of the Box2D MLP rollout: 0 = by population size, else <= 64 / lanes per env
    int tune_lander_per_wave;      // offspring per wave of the lockstep lander rollout: 0 = by population size, 1 / 2 / 4
    int tune_comm_p2p_timeout_ms;  // how long a peer-store exchange waits for a peer (0 = the default, 60 s)
    int tune_comm_p2p_keep_going;  // 1: exchanges continue after a time-out (the host polls ses_comm_p2p_status and recovers)
    int tune_fused_fitness;        // 1 (default): in a sharded ses_run_generations the FITNESS exchange needs no launch either (see fit_gv below)
    int tune_comm_granules_enabled; // 0: this handle's transport refuses granule exchanges (comm_p2p_granules_begin: unsupported) -- set by a host
                                    // whose check of them failed (ses/parallel.py); the flag-based exchanges carry everything then
    int tune_comm_granules;        // 1: ses_allgather_fitness over the peer-store transport moves {sequence, value} granules (no flag, no fence)
                                   // while a shard fits half a mailbox section; 0 (default): the kernel with sequence words -- for a whole
                                   // shard the per-float stores and polls cost more than the one release / acquire round they save
    int tune_openai_granules;      // 0: the shard form all-gathers its chunk partials as floats with a launch of its own also on the
                                   // peer-store transport (default 1: {sequence, value} granules stored by the gradient kernel itself)
    int tune_openai_sharded_tail;  // 0: ses_openai_sharded_ok says no (sharded runs use the replicated openai_es tail; A/B runs)
    int tune_openai_sharded_min_rows; // populations below this many rows IN TOTAL keep the replicated tail (default 8192)
};

namespace ses {

// the constant-rate (100 MHz) real-time counter shared by the whole GPU: timestamps taken inside kernels are
// comparable across kernels, streams and handles
__device__ __forceinline__ unsigned long long real_time() { return wall_clock64(); }

int set_error(int code, const char *fmt, ...);

#define SES_HIP_TRY(expr)                                                                          \
    do {                                                                                           \
        hipError_t e_ = (expr);                                                                    \
        if (e_ != hipSuccess)                                                                      \
            return ::ses::set_error(SES_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), \
                                    __FILE__, __LINE__);                                           \
    } while (0)

#define SES_REQUIRE(cond, ...)                                             \
    do {                                                                   \
        if (!(cond)) return ::ses::set_error(SES_ERR_INVALID_ARG, __VA_ARGS__); \
    } while (0)

inline int ceil_div(long long a, long long b) { return (int)((a + b - 1) / b); }

// A view of the peer-store mailboxes for kernels that exchange 8-byte {sequence, value} GRANULES themselves (the shard form
// of the openai_es tail: the gradient kernel stores its chunk partials straight into every rank's mailbox, the update
// kernel polls them -- the data is the flag, no exchange launch, no fence).  One aligned 8-byte store carries both words.
constexpr int P2P_GRANULE_MAX_WORLD = 16;
struct P2pGranuleView {
    unsigned long long *dst[P2P_GRANULE_MAX_WORLD];   // dst[r]: where THIS rank's granules go in rank r's mailbox (r = own rank included)
    const unsigned long long *src;                    // this rank's mailbox, the slot of this exchange: rank s's granules at src + s * section
    int section;                                      // granules per (slot, source rank) section
    int rank, world;
    uint32_t seq;                                     // the tag of this exchange
    unsigned long long timeout_ticks;
    uint32_t *err, *err_seen;                         // as k_allgather_p2p: host-visible mask, its copy in device memory
};
// reserves the next exchange of `comm`'s peer-store transport for a granule exchange of `granules` per rank; SES_ERR_UNSUPPORTED
// when the transport is not attached or a section cannot hold them, SES_ERR_COMM after an unrecovered time-out
int comm_p2p_granules_begin(ses_handle *comm, int granules, P2pGranuleView *view);

#if defined(__HIPCC__)
__device__ __forceinline__ void granule_store(unsigned long long *dst, uint32_t seq, uint32_t value_bits)
{
    // ONE aligned 8-byte store carries the value and the tag of the exchange it belongs to: whoever reads the tag it waits
    // for has the value (no flag, no fence); system scope: the mailbox may be another GPU's memory
    __hip_atomic_store(dst, ((unsigned long long)value_bits << 32) | (unsigned long long)seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// waits for the granule of exchange v.seq at `src` (a section of THIS rank's mailbox written by rank `from`): its value bits,
// or NaN after the time-out / when the word already carries a LATER exchange's tag (the peer gave up on this rank and moved
// on) -- with rank `from`'s bit set in the error words, as k_allgather_p2p does
__device__ __forceinline__ uint32_t granule_wait(const unsigned long long *src, const P2pGranuleView &v, int from)
{
    const unsigned long long t0 = real_time();
    unsigned long long limit = v.timeout_ticks;
    bool known_late = false;
    for (;;) {
        const unsigned long long g = __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        const uint32_t tag = (uint32_t)g;
        if (tag == v.seq) return (uint32_t)(g >> 32);
        if (!known_late) {                                            // (read once, on the slow path only)
            known_late = true;
            const uint32_t seen = __hip_atomic_load(v.err_seen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (((seen >> (from & 31)) & 1u) && limit > 200000ull) limit = 200000ull;      // 2 ms for a peer that was late before
        }
        if ((int32_t)(tag - v.seq) > 0 || real_time() - t0 > limit) {
            atomicOr_system(v.err, 1u << (from & 31));
            atomicOr(v.err_seen, 1u << (from & 31));
            return 0x7FC00000u;
        }
        __builtin_amdgcn_s_sleep(2);
    }
}

#endif

int openai_fused_fitness_ok(const ses_handle *h, int32_t n, int32_t per_rank, int32_t n_ranked);     // ses_strategy.hip
int elite_tail_small(ses_handle *h, const double *ep_return, int32_t n, int32_t k, const int32_t *parent_map, int32_t *alias_state,
                     int32_t *rank, float *fitness, float *best, int32_t *ids, int32_t *pidx, int32_t *alias,
                     unsigned long long *stamp, const float *parents, float sigma, uint64_t seed, uint64_t gen, float *mean_out);

int ensure_episode_scratch(ses_handle *h, size_t episodes);
int ensure_reduce_scratch(ses_handle *h, size_t bytes);
int comm_release(ses_handle *h);
void comm_p2p_set_timeout(ses_handle *h);


}  // namespace ses
