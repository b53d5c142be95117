// ses_generations.hip -- k whole generations in ONE call of the C ABI (ses_run_generations).
//
// The reference's generation loop (learning_strategies/evolution/loop.py:61-104) runs one `p.map(RolloutWorker, ...)`
// and one `strategy.evaluate(results)` per generation from Python.  On the device a generation of the reference's own
// configs (96-240 offspring) is 60-120 us of kernels, and the Python host needs ~96 us to enqueue it (ctypes calls,
// tensor bookkeeping): conf/cartpole.yaml and conf/simplespread.yaml ran at half the device's speed.  This file is the
// same sequence of entry points -- resets, fused rollout, episode mean, the strategy's tail, the next population --
// issued from C with the host-side scalars of the strategies (sigma decay, Adam's step scale, generation keys) advanced
// exactly as the Python classes advance them (double arithmetic, same libm pow / sqrt), so a run through this call is
// bit-identical to the per-generation path (tests/test_gpu_host_mirror.py).  No kernel lives here.
#include <cmath>
#include <cstring>

#include "ses_internal.h"

extern "C" {

int ses_run_generations(ses_handle *h, ses_gen_state *st, int32_t k, float *best, uint64_t *stamps)
{
    using namespace ses;
    SES_REQUIRE(h && st && best, "ses_run_generations: null argument");
    SES_REQUIRE(k >= 1, "ses_run_generations: k must be >= 1");
    SES_REQUIRE(st->strategy == SES_STRATEGY_OPENAI_ES || st->strategy == SES_STRATEGY_SIMPLE_EVOLUTION ||
                    st->strategy == SES_STRATEGY_SIMPLE_GENETIC, "ses_run_generations: unknown strategy %d", st->strategy);
    SES_REQUIRE(st->n >= 2 && (st->cur == 0 || st->cur == 1), "ses_run_generations: bad population size / buffer index");
    SES_REQUIRE(st->theta[0] && st->theta[1] && st->parents[0] && st->parents[1] && st->fitness && st->init,
                "ses_run_generations: null buffer");
    const bool openai = st->strategy == SES_STRATEGY_OPENAI_ES;
    if (openai) {
        SES_REQUIRE(st->adam_m[0] && st->adam_m[1] && st->adam_v[0] && st->adam_v[1], "ses_run_generations: openai_es needs the Adam buffers");
    } else {
        SES_REQUIRE(st->elite_num >= 1 && st->elite_num <= st->n && st->elite_num <= 1024 && st->parent_map && st->work_i32 &&
                        st->work_f32, "ses_run_generations: elite strategies need elite_num, parent_map and the work buffers");
        SES_REQUIRE(st->strategy != SES_STRATEGY_SIMPLE_EVOLUTION || st->alias_state, "ses_run_generations: simple_evolution needs alias_state");
    }
    // sharded run (one process per GPU): this rank rolls out its own rows, the fitness shards are all-gathered INSIDE the loop
    // (both transports of ses_allgather_fitness are plain stream enqueues), the strategy's tail follows
    const bool multi = st->world > 1;
    if (multi) {
        SES_REQUIRE(st->comm && st->fit_local, "ses_run_generations: a sharded run needs the transport handle and fit_local");
        SES_REQUIRE(st->per_rank >= 1 && (int64_t)st->per_rank * st->world >= st->n && st->first_row >= 0 &&
                        st->first_row % st->per_rank == 0 && st->first_row / st->per_rank < st->world,
                    "ses_run_generations: bad shard layout (%d ranks x %d rows for %d)", st->world, st->per_rank, st->n);
        const int64_t left = (int64_t)st->n - st->first_row;
        SES_REQUIRE(st->n_local == (int32_t)(left <= 0 ? 0 : left < st->per_rank ? left : st->per_rank),
                    "ses_run_generations: n_local %d is not this rank's share of %d rows", st->n_local, st->n);
        SES_REQUIRE(st->comm->stream == h->stream, "ses_run_generations: the transport handle must share the stream");
    }
    unsigned long long *const saved_stamp = h->stamp;
    const int n = st->n, ke = st->elite_num;
    const int n_loc = multi ? st->n_local : n;                         // rows of theta / init on this rank
    const int64_t first = multi ? st->first_row : 0;
    const bool sharded_tail = multi && openai && ses_openai_sharded_ok(h, st->comm, n, st->per_rank, st->world) == 1;
    // ... and then the fitness exchange itself needs no launch: the episode-mean kernel stores every value as a granule into every
    // rank's mailbox, the rank kernel of the tail polls the tiles it sorts (k_fitness_mean_granules, k_rank_sort_search<true>)
    // (up to 8192 rows the counting rank polls them, k_rank_count_granules: also where the tail runs replicated, e.g. 4096 rows in
    //  total over 8 ranks)
    const bool fused_fit = multi && openai && openai_fused_fitness_ok(h, n, st->per_rank, sharded_tail ? st->per_rank : n) == 1;   // (the slot size, not this rank's rows: a ragged last rank must decide like the others)
    // one GPU, openai_es, counting rank (up to 8192 rows): the episode mean is formed inside the rank count (k_rank_count_episodes)
    // -- ses_rollout leaves the per-episode returns, no mean kernel between the rollout and the tail
    const bool fused_mean = !multi && openai && h->tune_fused_mean && n <= 8192;
    // one GPU, elite strategies, up to 512 rows (the reference's own configs: 97 - 257): mean + rank + best + selection in ONE
    // launch, simple_evolution's elite rows + their mean in a second (elite_tail_small) instead of seven
    const bool fused_elite = !multi && !openai && h->tune_fused_elite && n <= 512;   // (one workgroup counts: 512 rows = 8 waves x 512 compares)
    int rc = SES_OK;
    // The env resets depend on (env seed, generation key) only: those of all k generations are drawn up front in ONE launch
    // (keyed like ESLoop._init_states), into a buffer the handle owns -- a 4 us kernel per generation less on the
    // critical path (the Python loop hides it on a side stream).  Above 64 MB the chunk is drawn generation by generation
    // into the caller's st->init instead.
    const int init_rows = st->shared_init ? 1 : (n_loc > 0 ? n_loc : 1);
    const size_t slice = (size_t)init_rows * h->cfg.eval_ep_num * st->init_width;
    const bool ahead = slice * (size_t)k * sizeof(float) <= (64u << 20);
    if (ahead) {
        if (h->gen_init_cap < slice * (size_t)k) {
            if (h->gen_init) { SES_HIP_TRY(hipStreamSynchronize(h->stream)); SES_HIP_TRY(hipFree(h->gen_init)); }
            h->gen_init = nullptr; h->gen_init_cap = 0;
            SES_HIP_TRY(hipMalloc(&h->gen_init, slice * (size_t)k * sizeof(float)));
            h->gen_init_cap = slice * (size_t)k;
        }
        rc = ses_init_states_uniform_gens(h, st->env_seed, st->pop_gen, k, st->shared_init ? 0 : first, init_rows, st->shared_init,
                                          st->init_width, st->init_lo, st->init_hi, h->gen_init);
    }
    for (int g = 0; g < k && rc == SES_OK; ++g) {
        const int cur = st->cur, nxt = cur ^ 1;
        const float *init = st->init;
        if (ahead) {
            init = h->gen_init + slice * (size_t)g;
        } else {
            rc = ses_init_states_uniform(h, st->env_seed, st->pop_gen, st->shared_init ? 0 : first, init_rows, st->shared_init,
                                         st->init_width, st->init_lo, st->init_hi, st->init);
            if (rc != SES_OK) break;
        }
        h->stamp = stamps ? (unsigned long long *)(stamps + 2 * g) : nullptr;          // end of the rollout phase
        P2pGranuleView fit_view;
        bool fused = false;
        if (fused_fit) {
            const int grc = comm_p2p_granules_begin(st->comm, st->per_rank, &fit_view);
            if (grc == SES_ERR_COMM) { rc = grc; break; }
            fused = grc == SES_OK;                                                      // (unsupported: RCCL only, or granules switched off)
            if (fused) { h->fit_gv = &fit_view; h->fit_own = st->fit_local; h->fit_per_rank = st->per_rank; }
        }
        unsigned long long *const rollout_stamp = h->stamp;
        if (fused_mean || fused_elite) { h->skip_mean = 1; h->stamp = nullptr; }
        if (n_loc > 0)
            rc = ses_rollout(h, st->theta[cur], init, st->shared_init ? 0 : 1, n_loc, st->mode, multi ? st->fit_local : st->fitness,
                             nullptr, nullptr);
        h->skip_mean = 0;
        if (rc != SES_OK) { h->fit_gv = nullptr; break; }
        if (fused_mean) { h->mean_src = h->ep_return; h->mean_stamp = rollout_stamp; }
        if (multi && !fused) {
            // loop.py:66-79, the gather half of Pool.map: fitness[r * per_rank + i] = rank r's fit_local[i] (a ragged last
            // shard ends in the -inf the caller put there once)
            rc = ses_allgather_fitness(st->comm, st->fit_local, st->per_rank, st->fitness);
            if (rc != SES_OK) break;
        }
        unsigned long long *const tail_stamp = stamps ? (unsigned long long *)(stamps + 2 * g + 1) : nullptr;
        if (openai) {
            // optimizers.py:43-47 via Adam.next_step_scale(); offspring_strategies.py _evaluate_fused
            st->adam_t += 1;
            const double t = (double)st->adam_t;
            const double a = st->learning_rate * std::sqrt(1.0 - std::pow(0.999, t)) / (1.0 - std::pow(0.99, t));
            const double sigma = st->sigma;
            st->sigma = st->sigma * st->sigma_decay;
            h->stamp = tail_stamp;
            if (sharded_tail)
                rc = ses_openai_generation_sharded(h, st->comm, st->fitness, n, st->seed, st->pop_gen, st->learning_rate, sigma, a,
                                                   st->parents[cur], st->adam_m[cur], st->adam_v[cur], st->parents[nxt],
                                                   st->adam_m[nxt], st->adam_v[nxt], (float)st->sigma, st->pop_gen + 1, first,
                                                   n_loc, st->per_rank, st->world, st->theta[nxt], best + g);
            else
                rc = ses_openai_generation(h, st->fitness, n, st->seed, st->pop_gen, st->learning_rate, sigma, a, st->parents[cur],
                                           st->adam_m[cur], st->adam_v[cur], st->parents[nxt], st->adam_m[nxt], st->adam_v[nxt],
                                           (float)st->sigma, st->pop_gen + 1, n_loc > 0 ? first : 0, n_loc, st->theta[nxt], best + g);
            st->pop_sigma = st->sigma;
            h->fit_gv = nullptr;
            h->fit_own = nullptr;
            h->mean_src = nullptr;
            h->mean_stamp = nullptr;
        } else {
            int32_t *rank = st->work_i32, *ids = rank + n, *pidx = ids + ke, *alias = pidx + ke;
            const bool evo = st->strategy == SES_STRATEGY_SIMPLE_EVOLUTION;
            if (fused_elite) {
                rc = elite_tail_small(h, h->ep_return, n, ke, st->parent_map, evo ? st->alias_state : nullptr, rank, st->fitness, best + g,
                                      ids, pidx, evo ? alias : nullptr, rollout_stamp, st->parents[cur], (float)st->pop_sigma, st->seed,
                                      st->pop_gen, evo ? st->parents[nxt] : nullptr);
            } else {
                rc = ses_rank_center(h, st->fitness, n, rank, nullptr, best + g);
                if (rc == SES_OK)
                    rc = ses_elite_select(h, rank, n, ke, st->parent_map, evo ? st->alias_state : nullptr, ids, pidx, evo ? alias : nullptr);
            }
            h->stamp = nullptr;                                                         // the elite rows are not "the next population"
            // the elite rows of the CURRENT population, rebuilt from (parents, parent map entry, row id): _select_elites
            float *rows = evo ? st->work_f32 : st->parents[nxt];
            if (rc == SES_OK && !(fused_elite && evo))
                rc = ses_perturb(h, st->parents[cur], pidx, ids, (float)st->pop_sigma, st->seed, st->pop_gen, 0, ke, rows);
            if (evo) {
                // mu = elite[0] = the reference's in-place elite sum (offspring_strategies.py:234-248), sigma decays BEFORE the
                // next population is drawn
                if (rc == SES_OK && !fused_elite) rc = ses_elite_mean(h, rows, alias, ke, st->parents[nxt]);
                st->sigma = st->sigma * st->sigma_decay;
                st->pop_sigma = st->sigma;
            } else {
                // simple_genetic: the elites are the parents; sigma decays AFTER regeneration (offspring_strategies.py:117-124)
                st->pop_sigma = st->sigma;
                st->sigma = st->sigma * st->sigma_decay;
            }
            h->stamp = tail_stamp;
            if (rc == SES_OK && n_loc > 0)
                rc = ses_perturb(h, st->parents[nxt], st->parent_map + first, nullptr, (float)st->pop_sigma, st->seed, st->pop_gen + 1,
                                 first, n_loc, st->theta[nxt]);
        }
        st->pop_gen += 1;
        st->cur = nxt;
    }
    h->stamp = saved_stamp;
    return rc;
}

}  // extern "C"
