/*
 * ses.h -- C ABI of libses_hip.so, the MI355X (gfx950) implementation of the simple-es
 * population rollout + fitness hot path.
 *
 * The reference (jinPrelude/simple-es) is pure Python and has no FFI of its own; its seam is
 *     results = p.map(RolloutWorker, arguments)                 learning_strategies/evolution/loop.py:66-79
 *     offsprings, best, sigma = strategy.evaluate(results)      learning_strategies/evolution/loop.py:82-84
 * Every entry point below names the reference code it replaces.  The Python host
 * (simple-es_amd/ses/_lib.py) binds them with ctypes; INTEGRATION.md shows the stub a reference
 * maintainer would add.
 *
 * Conventions
 *   - plain C: pointers + sizes, no torch / C++ types.
 *   - every array argument is a CALLER-OWNED DEVICE pointer (e.g. torch tensor .data_ptr());
 *     the library never frees or reallocates it.  The handle owns only scratch.
 *   - calls enqueue work on the handle's HIP stream and return immediately; ses_sync() blocks.
 *   - a handle is bound to one device and one stream and is NOT thread-safe: use it from one host thread at a
 *     time (one handle per rank / per stream).  Calls may allocate handle-owned scratch on first use or when a
 *     larger population arrives, so they are not hipGraph-capturable until the sizes have been seen once.
 *   - return value: SES_OK (0) or a negative SES_ERR_*; ses_last_error() gives a message
 *     (thread-local).  No C++ exception crosses the boundary.
 *   - "row" = one offspring's flat parameter vector, float32[P], in torch parameters() order
 *     (networks/neural_network.py:46-56): fc1.weight(32,S) fc1.bias(32)
 *     [gru.weight_ih(96,32) gru.weight_hh(96,32) gru.bias_ih(96) gru.bias_hh(96)]
 *     fc2.weight(A,32) fc2.bias(A).
 *   - results are index-ordered like Pool.map: fitness[i] belongs to row i.
 */
#ifndef SES_H_
#define SES_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SES_OK 0
#define SES_ERR_INVALID_ARG (-1)
#define SES_ERR_UNSUPPORTED (-2)
#define SES_ERR_HIP (-3)
#define SES_ERR_NO_DEVICE (-4)
#define SES_ERR_COMM (-5) /* RCCL not loadable, or a communicator call failed */

#define SES_HIDDEN 32 /* hidden width, hard-coded in networks/neural_network.py:12-17 */

/* env_id */
#define SES_ENV_NONE (-1)  /* no env: handle only serves policy-forward / strategy kernels */
#define SES_ENV_CARTPOLE 0 /* CartPole-v1 via envs/gym_wrapper.py:7-54 (conf/cartpole.yaml) */
#define SES_ENV_LUNARLANDER 1 /* LunarLanderContinuous-v2 via envs/gym_wrapper.py (conf/lunarlander_openai.yaml): gym's env     */
                              /* restated on a Box2D-style world (csrc/ses_lander.h, ses_b2.h + ses_b2_toi.h: discrete solver  */
                              /* and time-of-impact sub-stepping against the terrain; parity with gym / Box2D is              */
                              /* UNPINNED, neither is in the reference tree); num_state 8, num_action 4, continuous;        */
                              /* pomdp zeroes obs 2,3,5 (LunarLanderPOMDP, gym_wrapper.py:57-66); init rows: 16 uniforms   */
#define SES_ENV_SIMPLE_SPREAD 2 /* pettingzoo MPE simple_spread via envs/pettingzoo_wrapper.py:6-64 (conf/simplespread.yaml); */
                                /* num_state = 6*n_agents, num_action = 5, discrete, MLP policy shared by the agents      */
#define SES_ENV_BIPEDALWALKER 3 /* BipedalWalker-v3 via envs/gym_wrapper.py (conf/bipedalwalker.yaml): gym's env restated on   */
                                /* the same Box2D-style world (csrc/ses_walker.h; parity UNPINNED); num_state 24, num_action  */
                                /* 4, continuous, MLP policy; init rows: 4 floats (force uniform, 2 terrain-key words, pad)   */

/* rollout / env-step mode */
#define SES_MODE_EPISODIC 0     /* an env stops at done (reference semantics, loop.py:116)      */
#define SES_MODE_FIXED_LENGTH 1 /* synthetic-benchmark mode: physics keeps stepping after done, */
                                /* rewards gated by the alive flag; identical returns            */

typedef struct ses_handle ses_handle;

typedef struct ses_config {
    int32_t env_id;          /* SES_ENV_*                                                        */
    int32_t num_state;       /* network.num_state   (conf/cartpole.yaml:7)                       */
    int32_t num_action;      /* network.num_action  (conf/cartpole.yaml:8)                       */
    int32_t discrete_action; /* network.discrete_action                                          */
    int32_t gru;             /* network.gru                                                      */
    int32_t pomdp;           /* env.pomdp: CartPolePOMDP zeroes obs[1], obs[3] (gym_wrapper.py:69-77) */
    int32_t max_step;        /* env.max_step (gym_wrapper.py:37-39); CartPole-v1 TimeLimit = 500 */
    int32_t eval_ep_num;     /* --eval-ep-num, run_es.py:33-38                                   */
    int32_t device;          /* HIP device ordinal                                               */
    int32_t lanes_per_env;   /* 0 = choose from the population size; else 1, 2, 4, 8, 16 or 32   */
    int32_t n_agents;        /* simple_spread: agents (= landmarks) per env, 2 (reference) or 3; else 1 */
    int32_t physics64;       /* CartPole rollouts only: 1 = gym-order float64 dynamics (csrc/ses_cartpole.h),   */
                             /* 0 = folded-constant fp32 (default, benchmark path)                              */
} ses_config;

/* ---- lifecycle --------------------------------------------------------------------------- */
/* stream: a hipStream_t (e.g. torch.cuda.current_stream().cuda_stream) or NULL for the
 * device's default stream. */
int ses_create(const ses_config *cfg, void *stream, ses_handle **out);
int ses_destroy(ses_handle *h);
int ses_sync(ses_handle *h);
/* Development / test hook: which of the (result-identical) rollout kernels a handle picks.  Every kernel evaluates the
 * same canonical arithmetic, so no setting changes a result; the defaults are the measured crossovers.  Knobs:
 * "gru_ep_parallel_max" (default 4096), "gru_mfma_min_e" (12), "gru_mfma4_min_e" (7; 0 = never | 1 ... 8: eval_ep_num from which -- up to 8 -- the CartPole GRU rollout takes its policy step on v_mfma_f32_4x4x1_16b_f32), "gru_sequential" (0), "rollout_mix" (1),
 * "rollout_waves8" (1024: light waves of the mixed CartPole MLP split), "rollout_mix_light" (their lanes per env: 0 = choose | 8 | 16), "rollout_lpe32_max_envs" (0: CartPole MLP populations of up to this many envs run at 32 lanes per env), "rollout_mix_8_16" (default 1: populations of 8193 ... ~12 000 envs run 8 lanes per env on every SIMD and the rest at 16), "rollout_packed" (-1: populations of at most one wave per SIMD run the packed form of the CartPole MLP step | 0: never | 1: whenever 8 or 16 lanes share an env), "rollout_block" (64 | 256), "lander_offspring_per_wave" (0 = by population size | 1 | 2 | 4),
 * "box2d_lanes_per_env" (0 = by population size | 1 | 2 | ... | 64: lanes that share one env in the LunarLander / BipedalWalker MLP rollout),
 * "box2d_envs_per_wave" (0 = by population size | 1 ... 64 / lanes per env: different envs a wave of that rollout carries),
 * "env_step_block" (64 | 128 | 256), "env_step_waves_per_cu" (1 ... 32, default 7) and "env_step_lds_bytes" (-1 ... 65536,
 * default -1): workgroup size of ses_env_step's kernel and the LDS each workgroup reserves without touching it -- -1 derives
 * the reservation from the device's LDS per CU so that env_step_waves_per_cu waves stay in flight (7 is what the memory
 * system wants; ses_env_step_shape reports what the occupancy calculator makes of it), >= 0 is taken as given,
 * "es_final_max_chunks" (0: ses_openai_generation applies Adam in its own small launch; k > 0: inside the gradient kernel for
 * populations of up to k * 1024 rows), "comm_force_rccl" (1: ses_allgather_fitness uses the RCCL communicator although the
 * peer-store transport is attached -- for measuring one against the other), "comm_p2p_timeout_ms" (how long a peer-store
 * exchange waits for a peer's shard; 0 = default 60000), "comm_p2p_keep_going" (1: after a time-out later exchanges still
 * run instead of failing; the host polls ses_comm_p2p_status, agrees with the other ranks and rolls back -- ESLoop.run()),
 * "openai_sharded_tail" (default 1; 0: ses_openai_sharded_ok answers no, sharded runs keep the replicated openai_es tail),
 * "openai_sharded_min_rows" (default 8192: populations of fewer rows IN TOTAL keep the replicated tail -- the shard form costs a
 * second exchange and measured slower at 4096 rows in total),
 * "openai_granule_exchange" (default 1; 0: ses_openai_generation_sharded all-gathers its chunk partials with a launch of their
 * own also on the peer-store transport, as it does over RCCL -- for measuring one against the other),
 * "comm_granule_allgather" (default 0; 1: ses_allgather_fitness over the peer-store transport moves 8-byte {exchange number,
 * value} granules -- the data is its own flag -- while the shard fits half a mailbox section.  Measured slower for a whole
 * fitness shard, 12.4 us against 6.3 at 4096 floats; it is how a rank that does nothing else answers the granule exchange of
 * ses_openai_generation_sharded, tools/time_tail.py), "comm_granules_enabled" (default 1; 0: the handle's transport refuses
 * granule exchanges, the shard form of the tail then all-gathers its partials as floats), "fused_fitness_exchange" (default 1:
 * inside a sharded ses_run_generations above 8192 rows the fitness exchange needs no launch either -- the episode-mean kernel
 * stores every value as a granule into every rank's mailbox, the rank kernel polls the tiles it sorts; 0: ses_allgather_fitness
 * between rollout and tail), "fused_episode_mean" (default 1: ses_run_generations on one GPU, openai_es up to 8192 rows -- the
 * counting rank forms the episode means itself from the rollout's per-episode returns, no episode-mean launch; 0: as two launches),
 * "fused_elite_tail" (default 1: ses_run_generations on one GPU, simple_evolution / simple_genetic up to 512 rows -- episode mean,
 * rank, best reward and elite selection in one launch, simple_evolution's elite rows and their mean in a second; 0: seven launches),
 * "fused_apply_perturb" (default 1: ses_openai_generation, replicated form, policies up to 1024 parameters and populations up to
 * 16 384 rows -- every workgroup of the launch that writes the next population applies the Adam update itself; 0: a launch of its own).
 * The library itself reads no environment variable. */
int ses_set_tuning(ses_handle *h, const char *name, int32_t value);
/* Timing without events: from now on the last kernel of every ses_rollout (the episode mean: end of the rollout phase)
 * and every perturbation launch (ses_perturb, ses_perturb_host_noise, ses_openai_generation: the next population is
 * being written) of this handle stores the GPU's constant-rate real-time counter (100 MHz, common to all kernels,
 * streams and handles of the device) into *dst -- device-visible memory, e.g. pinned host memory the caller polls.
 * NULL switches it off.  Replaces the HIP events around `p.map(RolloutWorker, ...)` / `strategy.evaluate` that
 * loop.py:64-86 times with time.time(): an event costs the launch stream ~4.7 us, a stamp nothing. */
int ses_set_stamp(ses_handle *h, uint64_t *dst);
const char *ses_last_error(void);
const char *ses_version(void);
/* number of parameters P of GymEnvModel(S, A, _, gru)  (networks/neural_network.py:9-18) */
int ses_param_count(int32_t num_state, int32_t num_action, int32_t gru);
/* number of visible HIP devices, or a negative error (never initialises a context) */
int ses_device_count(void);

/* ---- K1: offspring perturbation ---------------------------------------------------------- */
/*
 * theta[i,:] = parents[k,:] + sigma * eps(seed, gen, row_i, :)   if parent_idx[i] = k >= 0
 * theta[i,:] = parents[k,:]                                      if parent_idx[i] = -1-k
 * row_i = row_ids ? row_ids[i] : first_row + i   (GLOBAL offspring index: the noise is a pure
 * function of (seed, gen, row, column), so any shard of any GPU count reproduces the same rows).
 * eps: rocRAND Philox4x32-10 device engine + deterministic Box-Muller.
 * Replaces the deepcopy + np.random.normal loops of offspring_strategies.py:53-60 (simple_genetic),
 * :169-176 (simple_evolution), :312-326 (openai_es).  parent_idx == NULL: every row perturbs parent 0.
 */
int ses_perturb(ses_handle *h, const float *parents, const int32_t *parent_idx, const int32_t *row_ids,
                float sigma, uint64_t seed, uint64_t gen, int64_t first_row, int32_t n_rows, float *theta);
/* the raw normals eps[n_rows, P] of the same stream (tests, diagnostics) */
int ses_noise(ses_handle *h, uint64_t seed, uint64_t gen, int64_t first_row, int32_t n_rows, float *eps);
/*
 * Reference-stream mode: eps64[n_rows,P] are float64 normals drawn on the host from the
 * reference's own generator (np.random.normal, offspring_strategies.py:57,173,320) and uploaded.
 *   theta[i,:]        = (float)( (double)parent + eps64 * sigma )   -- offspring_strategies.py:322
 *   eps_store[i,:]    = (float)( (double)parent + eps64 )           -- :321 (openai_es.epsilons quirk,
 *                        SURVEY 3.4-4); may be NULL
 * bit-identical to the reference's float64-then-float32 arithmetic.  np.random.normal(0, sigma) of
 * :57/:173 is sigma * z on the same gauss stream, so the host always uploads standard normals z.
 */
int ses_perturb_host_noise(ses_handle *h, const float *parents, const int32_t *parent_idx,
                           const double *eps64, double sigma, int32_t n_rows, float *theta,
                           float *eps_store);
/* Reset distribution U(lo,hi)^width from the ENV_INIT Philox stream, out[n_rows,E,width]
 * (CartPole: width 4, U(-0.05,0.05); simple_spread: width 4*n_agents = agent then landmark positions,
 * U(-1,1)).  shared != 0 keys every row as offspring 0 (common random numbers).  The reference never
 * seeds its env: initial states are an explicit input of this library. */
int ses_init_states_uniform(ses_handle *h, uint64_t seed, uint64_t gen, int64_t first_row, int32_t n_rows,
                            int32_t shared, int32_t width, float lo, float hi, float *out);
/* The same for `gens` consecutive generations in ONE launch: out[g, n_rows, E, width] = what ses_init_states_uniform writes
 * for generation gen0 + g.  The resets depend on (seed, generation, row) only, so a generation loop draws them a chunk
 * ahead instead of once per generation (ESLoop.rollout, ses_run_generations). */
int ses_init_states_uniform_gens(ses_handle *h, uint64_t seed, uint64_t gen0, int32_t gens, int64_t first_row,
                                 int32_t n_rows, int32_t shared, int32_t width, float lo, float hi, float *out);

/* ---- K2: population-batched policy forward (networks/neural_network.py:20-36) ------------- */
/* n independent (row, observation[, hidden]) triples.  hidden: float32[n,32] in/out, NULL for MLP.
 * logits[n,A]: fc2 pre-activation; act[n,A] (may be NULL): tanh(logits), the continuous action;
 * action[n]: argmax(logits), first maximum wins (== torch.argmax(softmax) up to fp ties). */
int ses_policy_forward(ses_handle *h, const float *theta, const float *obs, float *hidden, int32_t n,
                       float *logits, float *act, int32_t *action);

/* ---- K3: SoA env step (envs/gym_wrapper.py:32-45 around the third-party gym physics) ------- */
/* CartPole, one lane = one env, fp32 state in structure-of-arrays form.
 * status[i]: bits 0..30 steps taken, bit 31 done.  ret[i] += 1 per live step.
 * Algorithmic traffic 52 B / env-step: loads x,xd,th,thd,action,ret,status; stores all but action.
 * States are saturated at |x|, |xd|, |thd| <= 1e4 and |th| <= 0.75 rad: past-terminal values only (an episode ends
 * at |x| > 2.4 or |th| > 0.2095), which matter in SES_MODE_FIXED_LENGTH where finished envs keep stepping. */
int ses_env_step(ses_handle *h, int32_t n, int32_t mode, float *x, float *xd, float *th, float *thd,
                 const int32_t *action, float *ret, uint32_t *status);

/* ---- step-wise env entry for every env (envs/gym_wrapper.py:23-45, envs/pettingzoo_wrapper.py:22-58) ------------------ */
/* `env.reset()` / `env.step(action)` of the reference's wrappers for n independent envs, one lane = one env, through the
 * SAME device functions the fused rollouts call (csrc/ses_envs.hip).  The state of an env is an opaque blob of
 * ses_env_state_bytes(h) bytes in caller-owned device memory (CartPole 16 B; simple_spread 24 * n_agents + 4; LunarLander /
 * BipedalWalker: the Box2D-style world of the env followed by the episode's terrain heights).
 *   ses_env_reset:  init[n, W] (W as for ses_rollout: CartPole 4, simple_spread 4 * n_agents, LunarLander 16, BipedalWalker 4)
 *                   -> state[n], obs[n, ses_env_obs_width(h)]  (simple_spread: [n, n_agents, 6 * n_agents]); the Box2D envs
 *                   end their reset with gym's no-op step.
 *   ses_env_step_generic: action = int32[n] (CartPole), int32[n, n_agents] (simple_spread) or float32[n, num_action]
 *                   (LunarLander uses components 0 and 1; BipedalWalker all four), already in the env's
 *                   action space (the policy's tanh output) -> obs, reward[n] (simple_spread: the team reward of the cycle,
 *                   pettingzoo_wrapper.py:45-52), done[n] = the ENV's own termination (CartPole: |x| > 2.4 or |th| > 12
 *                   deg; simple_spread: after 25 cycles; Box2D: crash / out of bounds / asleep).  Truncation at env.max_step is
 *                   the wrapper's (gym_wrapper.py:37-39).  POMDP handles zero the masked observation components
 *                   (gym_wrapper.py:57-77).  A finished env may be stepped on (its state keeps evolving); callers reset it. */
int ses_env_state_bytes(ses_handle *h);
int ses_env_obs_width(ses_handle *h);
int ses_env_reset(ses_handle *h, const float *init, int32_t n, void *state, float *obs);
int ses_env_step_generic(ses_handle *h, void *state, const void *action, int32_t n, float *obs, float *reward,
                         int32_t *done);

/* The launch shape ses_env_step (and ses_stream_probe) use on this device with the current knobs: threads per workgroup, the
 * LDS bytes each workgroup reserves, the waves per CU the HIP occupancy calculator gives that shape (0: unknown), and the
 * device's LDS per CU the default reservation is derived from.  Any pointer may be NULL. */
int ses_env_step_shape(ses_handle *h, int32_t *block, int32_t *lds_bytes, int32_t *waves_per_cu, int32_t *lds_per_cu);

/* Measurement aid for the roofline of ses_env_step (no reference counterpart): the same 13 streams -- 7 x 16-byte
 * non-temporal loads and 6 x 16-byte non-temporal stores per lane over the same arrays, same grid -- with no arithmetic
 * in between; every value is written back unchanged.  Its duration is what the memory system of the box gives this
 * access pattern; bench.py prints it next to the env-step kernel's (roofline.copy13_*).  n a multiple of 4, arrays
 * 16-byte aligned. */
int ses_stream_probe(ses_handle *h, int32_t n, float *x, float *xd, float *th, float *thd, const int32_t *action,
                     float *ret, uint32_t *status);

/* ---- fused rollout: RolloutWorker for the whole shard (loop.py:108-125) -------------------- */
/*
 * theta[n_rows,P]; init: float32 [E,W] (init_per_offspring = 0, shared) or [n_rows,E,W], W = 4 for
 * CartPole (the state), 4*n_agents for simple_spread (agent positions, landmark positions).
 * fitness[n_rows] = sum over the E episodes of the undiscounted return / E  (loop.py:124).
 * ep_return (float64[n_rows,E]) and ep_steps (int32[n_rows,E]) may be NULL.
 * The whole episode loop (policy forward + env step, <= max_step iterations) runs inside one kernel
 * with env state, GRU state and the offspring's weights held in registers.
 */
int ses_rollout(ses_handle *h, const float *theta, const float *init, int32_t init_per_offspring,
                int32_t n_rows, int32_t mode, float *fitness, double *ep_return, int32_t *ep_steps);

/* ---- K4: rank-centred fitness shaping (offspring_strategies.py:380-398) --------------------- */
/* rank[i] = number of offspring that beat i (reward descending; ties: higher index first, i.e.
 * np.flip(np.argsort(kind="stable"))).  weights[i] = ((n-1-rank)/(n-1) - 0.5) / std, float64,
 * with the closed-form std sqrt((n+1)/(12(n-1))) of the rank grid.  weights may be NULL.
 * best (float32[1], may be NULL) receives the fitness of the offspring with rank 0, i.e. max(rewards): the
 * `best_reward` that evaluate() returns (offspring_strategies.py:420-434), so that the host reads one float back
 * instead of reducing the vector. */
int ses_rank_center(ses_handle *h, const float *fitness, int32_t n, int32_t *rank, double *weights, float *best);

/* ---- K5: ES gradient + Adam (offspring_strategies.py:400-416, optimizers.py:13-24,42-57) ---- */
/* One generation's fitness loop of openai_es (offspring_strategies.py:380-419 followed by _gen_offsprings :284-328) in
 * three launches (up to 8192 offspring): counting rank with the keys formed inline, ES-gradient partials with the
 * rank-centring weights formed inline and Adam applied by the last workgroup, Philox perturbation of the next population
 * (larger populations: keys, tile sort, search, gradient, update, perturbation).  Bit-identical to ses_rank_center +
 * ses_es_update_philox(skip_row0 = 1) + ses_perturb called one after the other (tests/test_gpu_host_mirror.py).
 *   fitness[n]: the gathered fitness of the evaluated population (noise generation `gen`, std `sigma`);
 *   (mu, m, v)_in -> (mu, m, v)_out: distinct float32[P] buffers, the caller ping-pongs them;
 *   theta_next[n_rows, P]: rows [first_row, first_row + n_rows) of the next population (noise generation next_gen, std
 *   next_sigma; global row 0 = the new mu itself); best: optional float32[1] <- max(fitness). */
int ses_openai_generation(ses_handle *h, const float *fitness, int32_t n, uint64_t seed, uint64_t gen, double lr,
                          double sigma, double adam_a, const float *mu_in, const float *m_in, const float *v_in,
                          float *mu_out, float *m_out, float *v_out, float next_sigma, uint64_t next_gen,
                          int64_t first_row, int32_t n_rows, float *theta_next, float *best);
/* The same generation when the population is sharded over `world` ranks (loop.py:66-84 with one process per GPU) WITHOUT
 * every rank repeating the O(n) work: this rank ranks only its own rows [first_row, first_row + n_rows) against the gathered
 * fitness (n <= 8192: counting rank; above: every workgroup sorts one 1024-key tile and searches it for 1024 own rows) and
 * accumulates the ES gradient over its own 1024-row chunks only; the ranks then exchange their [chunks per rank, P] chunk
 * partials -- with the candidates for max(fitness) behind them -- through `comm`'s transport (58 KB in total at 65 536 x 226)
 * and the unchanged ordered update adds the chunks in ascending order.  On the peer-store transport the exchange needs no
 * launch: the gradient kernel stores every partial as one 8-byte {exchange number, value} granule straight into every rank's
 * mailbox and the update kernel polls the granules where they land (the data is the flag); over RCCL, or when a mailbox
 * section cannot hold the granules, the partials are all-gathered as floats by ses_allgather_fitness in between.
 * Bit-identical to ses_openai_generation for ANY world size, because a rank's slot of per_rank = ceil(n / world) rows is a
 * whole number of the gradient's 1024-row chunks -- that is the condition; ses_openai_sharded_ok tells (1 / 0) whether it
 * holds and `comm` (a handle on the same device and stream that owns a transport of `world` ranks which takes the payload)
 * can carry it.  Otherwise: SES_ERR_UNSUPPORTED, use ses_openai_generation.  Collective: every rank calls it. */
int ses_openai_sharded_ok(ses_handle *h, ses_handle *comm, int32_t n, int32_t per_rank, int32_t world);
int ses_openai_generation_sharded(ses_handle *h, ses_handle *comm, const float *fitness, int32_t n, uint64_t seed, uint64_t gen,
                                  double lr, double sigma, double adam_a, const float *mu_in, const float *m_in,
                                  const float *v_in, float *mu_out, float *m_out, float *v_out, float next_sigma,
                                  uint64_t next_gen, int64_t first_row, int32_t n_rows, int32_t per_rank, int32_t world,
                                  float *theta_next, float *best);
/*
 * grad = (-lr / (n*sigma)) * sum_i weights[i] * eps_i ;  Adam (beta1 = 0.99, beta2 = 0.999, eps = 1e-8)
 * with step scale adam_a = lr*sqrt(1-beta2^t)/(1-beta1^t) computed by the host; mu, m, v updated in place.
 * _philox: eps_i regenerated from (seed, gen, row i) for ALL n rows -- every rank of a multi-GPU job
 *          computes the identical update without a second collective.  Row 0 has eps = 0
 *          (offspring_strategies.py:302-310) when skip_row0 != 0.
 * _stored: eps_store[n,P] from ses_perturb_host_noise, accumulated sequentially in the reference's
 *          order and precision (float32 accumulator, float64 products) -- bit-exact mirror.
 * grad_out (float32[P]) may be NULL.
 */
int ses_es_update_philox(ses_handle *h, const double *weights, int32_t n, int32_t skip_row0, uint64_t seed,
                         uint64_t gen, double lr, double sigma, double adam_a, float *mu, float *m, float *v,
                         float *grad_out);
int ses_es_update_stored(ses_handle *h, const double *weights, int32_t n, const float *eps_store, double lr,
                         double sigma, double adam_a, float *mu, float *m, float *v, float *grad_out);

/* ---- K6: elite selection + mean (offspring_strategies.py:112-116, 234-248) ------------------ */
/* elite_ids[j] = index of the offspring with rank j, j < k. */
int ses_elite_ids(ses_handle *h, const int32_t *rank, int32_t n, int32_t k, int32_t *elite_ids);
/* The elite bookkeeping of one generation in a single launch, nothing read back by the host:
 *   elite_ids[j]        = index of the offspring with rank j (as ses_elite_ids);
 *   elite_parent_idx[j] = parent_map[elite_ids[j]], the entry of the current population's parent map -- what
 *                         ses_perturb(row_ids = elite_ids, parent_idx = elite_parent_idx) needs to rebuild the rows;
 *   alias_first[j]      = the flags of ses_elite_mean for simple_evolution (offspring_strategies.py:234-248 sums the
 *                         elites into elite 0 in place; while population slots 0 and 1 are the same module object,
 *                         an elite in the other of the two slots doubles the running sum): set when *alias_state != 0,
 *                         elite_ids[0] is 0 or 1, and elite_ids[j] is the other of 0 / 1;
 *   *alias_state        is then updated: elite_ids[0] == 0, or elite_ids[0] == 1 while the state was set.
 * alias_state and alias_first may be NULL together (simple_genetic).  k <= 1024. */
int ses_elite_select(ses_handle *h, const int32_t *rank, int32_t n, int32_t k, const int32_t *parent_map,
                     int32_t *alias_state, int32_t *elite_ids, int32_t *elite_parent_idx, int32_t *alias_first);
/* rows[k,P] (the elites, best first) -> mean[P] = ((rows[0] + rows[1]) + ... ) / k in float32, the
 * reference's in-place order (:241-248).  alias_first[j] != 0 (j >= 1) marks an elite that is the same
 * module object as elite 0, for which the reference's `mu += elite` doubles the running sum
 * (SURVEY 3.4-6); NULL = no aliasing. */
int ses_elite_mean(ses_handle *h, const float *rows, const int32_t *alias_first, int32_t k, float *mean);
/* dst[i,:] = src[ids[i],:] */
int ses_gather_rows(ses_handle *h, const float *src, const int32_t *ids, int32_t n_ids, float *dst);

/* ---- k whole generations per call (learning_strategies/evolution/loop.py:61-104 from C) ------------------------------------ */
/*
 * The generation loop of ESLoop.run() -- env resets, fused rollout, episode mean, strategy.evaluate, next population -- for
 * k generations in ONE call: the same entry points as above issued back to back from C, with the strategies' host-side
 * scalars (sigma decay, Adam's step scale, generation keys) advanced exactly as the Python classes advance them, so the
 * results are bit-identical to k per-generation calls.  Why: a generation of the reference's own configs (96-240
 * offspring) is 60-120 us of kernels and took ~96 us of Python to enqueue; from C the device is the limit.
 * Counter-based noise only.  Everything is enqueued, nothing is waited for.
 * Sharded runs (world > 1, one process per GPU): the rollout covers this rank's n_local rows, ses_allgather_fitness on `comm`
 * (the handle that owns the transports; same device and stream) gathers the shards inside the loop, openai_es continues
 * with ses_openai_generation_sharded where ses_openai_sharded_ok allows it, everything else with the replicated tail over
 * all n rows.  A peer-store time-out is the caller's to poll (ses_comm_p2p_status) at its own boundaries.
 *
 * ses_gen_state: the caller fills the fixed part and the buffers once, the call advances the rest in place:
 *   strategy            SES_STRATEGY_*
 *   n                   population rows: openai_es offspring_num (row 0 = mu); simple_evolution offspring_num + 1
 *                       ([mu, elite0, children]); simple_genetic elite_num * (offspring_num / elite_num)
 *   theta[2], parents[2], adam_m[2], adam_v[2]   ping-pong halves, `cur` says which holds the current generation's;
 *                       parents: mu[P] (openai_es, simple_evolution) or elites[elite_num, P] (simple_genetic)
 *   parent_map          int32[n] on the device, the strategy's constant map as for ses_perturb (elite strategies)
 *   alias_state         int32[1] on the device (simple_evolution, see ses_elite_select)
 *   fitness[n] (sharded: [world * per_rank], the gathered vector), init[(shared_init ? 1 : rows of theta) * E * init_width],
 *   work_i32[n + 3 * elite_num], work_f32[elite_num * P]
 *   world, first_row, n_local, per_rank, comm, fit_local   sharded runs only (world <= 1: ignored): this rank owns the global
 *                       rows [first_row, first_row + n_local) with first_row = rank * per_rank, per_rank = ceil(n / world);
 *                       theta[] then has n_local rows; fit_local[per_rank] receives this rank's fitness, its tail
 *                       [n_local, per_rank) holds -inf, written once by the caller
 *   sigma / pop_sigma   curr_sigma of the strategy / the sigma the current population was drawn with
 *   pop_gen             generation key of the current population (its noise and its env resets)
 *   adam_t              Adam's step counter
 * best: float[k], device or PINNED HOST memory (the kernels store straight into it) <- max(fitness) of each generation;
 * stamps: optional uint64[k][2] in device-visible memory <- the GPU's 100 MHz counter at the end of each rollout phase and
 * at the start of the launch that writes the next population (ses_set_stamp).  The handle's own stamp is left as it was.
 */
#define SES_STRATEGY_OPENAI_ES 0        /* offspring_strategies.py:262-434 */
#define SES_STRATEGY_SIMPLE_EVOLUTION 1 /* offspring_strategies.py:128-259 */
#define SES_STRATEGY_SIMPLE_GENETIC 2   /* offspring_strategies.py:11-125  */
typedef struct ses_gen_state {
    int32_t strategy, n, elite_num, mode, shared_init, init_width;
    float init_lo, init_hi;
    uint64_t seed, env_seed;
    double learning_rate, sigma_decay;
    double sigma, pop_sigma;
    uint64_t pop_gen;
    int64_t adam_t;
    int32_t cur, world;
    float *theta[2];
    float *parents[2];
    float *adam_m[2];
    float *adam_v[2];
    int32_t *parent_map;
    int32_t *alias_state;
    float *fitness;
    float *init;
    int32_t *work_i32;
    float *work_f32;
    int64_t first_row;
    int32_t n_local, per_rank;
    ses_handle *comm;
    float *fit_local;
} ses_gen_state;
int ses_run_generations(ses_handle *h, ses_gen_state *st, int32_t k, float *best, uint64_t *stamps);

/* ---- multi-GPU: fitness all-gather over RCCL (replaces the gather half of Pool.map, loop.py:66-79) ---------- */
/*
 * The population shards over one process per GPU: rank r of W owns n_per_rank = ceil(N / W) consecutive global
 * rows (the last rank pads its shard, e.g. with -inf).  The ONE exchange of a generation is the all-gather of the
 * float32 fitness shards, N * 4 bytes (16 KB at N = 4096): latency-bound on xGMI, one ncclAllGather, no bucketing.
 * Everything after it (rank shaping, ES update, elite rows) is recomputed identically on every rank from the
 * counter-based noise, so there is no second collective.
 *   ses_comm_unique_id : rank 0 fills id[SES_COMM_ID_BYTES] (ncclGetUniqueId); the host hands the bytes to the other
 *                        ranks by whatever it has (a file, MPI, a torch.distributed store, ...).
 *   ses_comm_init      : collective over the W ranks; binds an RCCL communicator for the handle's device to the handle.
 *   ses_allgather_fitness : all[r * n_per_rank + i] = local_r[i] on every rank, enqueued on the handle's stream.
 * RCCL (librccl.so.1) is loaded on the first ses_comm_* call; single-GPU users never load it.
 */
#define SES_COMM_ID_BYTES 128
int ses_comm_unique_id(void *id);
int ses_comm_init(ses_handle *h, int32_t rank, int32_t world, const void *id);
/* rank / world of the handle's communicator (world = 0: none) and the RCCL version code; any pointer may be NULL */
int ses_comm_info(ses_handle *h, int32_t *rank, int32_t *world, int32_t *rccl_version);
int ses_comm_destroy(ses_handle *h); /* also done by ses_destroy */
int ses_allgather_fitness(ses_handle *h, const float *local, int32_t n_per_rank, float *all);
/*
 * The same exchange by PEER STORES inside one node (no RCCL, no ring): every rank owns a mailbox in fine-grained device
 * memory, the W - 1 peers map it (hipIpc*), and one small kernel per rank stores its shard into every peer's mailbox,
 * publishes a sequence number and collects the peers' shards from its own mailbox -- the latency of one xGMI store
 * instead of W - 1 ring hops behind a library launch.  Once attached, ses_allgather_fitness uses it for n_per_rank <=
 * max_per_rank (results identical: it is a copy).  A wait that exceeds the time-out ("comm_p2p_timeout_ms", default 60 s)
 * NaN-fills the missing shard and sets that peer's bit in a host-visible word: ses_comm_p2p_status reads it at any time
 * without touching the stream (once the kernels that consume an exchange have finished, the word is final for it -- check
 * it before a generation's results are used or checkpointed), and the next ses_allgather_fitness fails with SES_ERR_COMM
 * unless "comm_p2p_keep_going" is set; after ses_comm_p2p_detach the handle is back on RCCL.
 *   ses_comm_p2p_export : allocate this rank's mailbox; handle[SES_COMM_P2P_HANDLE_BYTES] is what the peers need.
 *   ses_comm_p2p_attach : handles = the W exported handles in rank order (W * SES_COMM_P2P_HANDLE_BYTES bytes, gathered
 *                         by the host's control plane); maps the peers.  Every rank must have exported before any attaches.
 *   ses_comm_p2p_info   : world (0 = not attached), max_per_rank, exchanges done so far; any pointer may be NULL.
 * Ranks may share a GPU (several processes on one device): that is how the single-GPU tests drive this path.
 */
#define SES_COMM_P2P_HANDLE_BYTES 64
int ses_comm_p2p_export(ses_handle *h, int32_t rank, int32_t world, int32_t max_per_rank, void *handle);
int ses_comm_p2p_attach(ses_handle *h, const void *handles);
/* The same transport between handles of ONE process (a host that drives several GPUs, or several streams of one GPU, from
 * one process; also how one test process forms worlds of 8 and 16 ranks on a single GPU): every handle exports as above, then
 * attaches the peers' HANDLES -- peers[world] in rank order, peers[rank] == h -- instead of their IPC bytes: nothing is
 * mapped, the peers' mailboxes are addressed directly, so they must live on the same device or on devices the caller has
 * enabled peer access between, and every handle must detach before any of them is destroyed.  An exchange kernel WAITS for
 * kernels of its peers: each handle needs not just a stream but a HARDWARE QUEUE of its own.  Handles on different devices
 * have that by construction; for handles that share a device it is the runtime's choice (GPU_MAX_HW_QUEUES, default 4, and
 * the order in which streams were created) -- two streams on one queue are a dead wait until the time-out -- unless the
 * streams come from ses_stream_create_exclusive below (tests/test_gpu_sharded_tail.py: one process, a stream per rank). */
int ses_comm_p2p_attach_local(ses_handle *h, ses_handle *const *peers);
/* A HIP stream with a hardware queue of its OWN on `device`, for handles that share a device and wait for each other's kernels
 * (above).  The runtime pools the queues of ordinary streams -- GPU_MAX_HW_QUEUES of them per priority, shared round-robin --
 * but a stream created with a CU mask (hipExtStreamCreateWithCUMask) never enters the pool: it always gets a queue created for
 * it.  The mask given here names every CU of the device, so it restricts nothing.  *stream is a hipStream_t for ses_create
 * (and torch.cuda.ExternalStream); ses_stream_destroy ends it.  This is what makes the in-process many-rank rig deterministic
 * (tests/test_gpu_sharded_tail.py); one process per GPU -- the product's way to run -- never needs it. */
int ses_stream_create_exclusive(int32_t device, void **stream);
int ses_stream_destroy(void *stream);
int ses_comm_p2p_info(ses_handle *h, int32_t *world, int32_t *max_per_rank, int32_t *exchanges);
/* exchanges issued over the attached transport so far, by kind: with sequence words (ses_allgather_fitness) / as granules (the
 * exchanges kernels do themselves: chunk partials and, inside ses_run_generations, the fitness; the granule all-gather) */
int ses_comm_p2p_counts(ses_handle *h, int32_t *flag_exchanges, int32_t *granule_exchanges);
int ses_comm_p2p_status(ses_handle *h, uint32_t *timed_out_mask);   /* bit r: an exchange gave up waiting for rank r */
/* Clears the time-out mask (drains the handle's stream first): for a host that has dealt with a failed exchange and whose ranks
 * have AGREED to go on using this transport -- e.g. after a failed check of the granule exchanges at attach time, which switches
 * those off ("comm_granules_enabled" = 0) and keeps the flag-based exchanges. */
int ses_comm_p2p_reset_status(ses_handle *h);
int ses_comm_p2p_detach(ses_handle *h);

#ifdef __cplusplus
}
#endif
#endif /* SES_H_ */
