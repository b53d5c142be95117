// ses_strategy.hip -- the fitness-loop side of the hot path (strategy.evaluate, loop.py:82-84):
//   K1 perturbation (Philox / host-noise), K4 rank-centring, K5 ES gradient + Adam, K6 elite mean.
// Reference: learning_strategies/evolution/offspring_strategies.py, learning_strategies/optimizers.py.
#include "ses_internal.h"
#include "ses_rng.h"

namespace ses {

// ------------------------------------------------------------------------------------------------ K1
// one thread = one Philox call = 4 consecutive parameters of one offspring
__global__ __launch_bounds__(256) void k_perturb(const float *__restrict__ parents,
                                                 const int32_t *__restrict__ parent_idx,
                                                 const int32_t *__restrict__ row_ids, float sigma, uint64_t seed,
                                                 uint64_t gen, long long first_row, int n_rows, int P, int quads,
                                                 float *__restrict__ theta, unsigned long long *__restrict__ stamp)
{
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (stamp && t == 0) *stamp = real_time();                           // ses_set_stamp: the next population is being written
    if (t >= (long long)n_rows * quads) return;
    const int i = (int)(t / quads);
    const int q = (int)(t - (long long)i * quads);
    const int32_t pi = parent_idx ? parent_idx[i] : 0;
    const float *src = parents + (size_t)(pi >= 0 ? pi : -1 - pi) * P + 4 * q;
    float *dst = theta + (size_t)i * P + 4 * q;
    const int lim = P - 4 * q < 4 ? P - 4 * q : 4;
    if (pi < 0) {
        for (int l = 0; l < lim; ++l) dst[l] = src[l];
        return;
    }
    const uint32_t row = (uint32_t)(row_ids ? (long long)row_ids[i] : first_row + i);
    float z[4];
    normal4(seed, gen, row, (uint32_t)q, z);
    for (int l = 0; l < lim; ++l) dst[l] = fma_(sigma, z[l], src[l]);
}

__global__ __launch_bounds__(256) void k_noise(uint64_t seed, uint64_t gen, long long first_row, int n_rows, int P,
                                               int quads, float *__restrict__ eps)
{
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (long long)n_rows * quads) return;
    const int i = (int)(t / quads);
    const int q = (int)(t - (long long)i * quads);
    float z[4];
    normal4(seed, gen, (uint32_t)(first_row + i), (uint32_t)q, z);
    const int lim = P - 4 * q < 4 ? P - 4 * q : 4;
    for (int l = 0; l < lim; ++l) eps[(size_t)i * P + 4 * q + l] = z[l];
}

// float64 host noise, the reference's own rounding: float32(float64(parent) + eps*sigma)
__global__ __launch_bounds__(256) void k_perturb_host_noise(const float *__restrict__ parents,
                                                            const int32_t *__restrict__ parent_idx,
                                                            const double *__restrict__ eps64, double sigma,
                                                            int n_rows, int P, float *__restrict__ theta,
                                                            float *__restrict__ eps_store,
                                                            unsigned long long *__restrict__ stamp)
{
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (stamp && t == 0) *stamp = real_time();
    if (t >= (long long)n_rows * P) return;
    const int i = (int)(t / P);
    const int p = (int)(t - (long long)i * P);
    const int32_t pi = parent_idx ? parent_idx[i] : 0;
    const float base = parents[(size_t)(pi >= 0 ? pi : -1 - pi) * P + p];
    if (pi < 0) {
        theta[t] = base;
        if (eps_store) eps_store[t] = base;
        return;
    }
    const double e = eps64[t];
    const double scaled = e * sigma;                       // offspring_strategies.py:322  epsilon * sigma
    theta[t] = (float)((double)base + scaled);             // in-place += on a float32 view
    if (eps_store) eps_store[t] = (float)((double)base + e);  // :321  eps_param += epsilon
}

__global__ __launch_bounds__(256) void k_init_states_uniform(uint64_t seed, uint64_t gen, long long first_row,
                                                             int n_rows, int E, int S, int shared, float lo,
                                                             float span, float *__restrict__ out)
{
    // blockIdx.y: consecutive generations (ses_run_generations draws the resets of a whole chunk of generations in one
    // launch: generation gen + y goes to out + y * n_rows * E * S)
    const int sq = (S + 3) / 4;
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (long long)n_rows * E * sq) return;
    const int q = (int)(t % sq);
    const int e = (int)((t / sq) % E);
    const int i = (int)(t / ((long long)sq * E));
    gen += blockIdx.y;
    out += (size_t)blockIdx.y * n_rows * E * S;
    const uint4 r = philox_words(seed, TAG_ENV_INIT, gen, shared ? 0u : (uint32_t)(first_row + i), (uint32_t)(e * 8 + q));
    const uint32_t w[4] = {r.x, r.y, r.z, r.w};
    for (int l = 0; l < 4 && 4 * q + l < S; ++l)
        out[((size_t)i * E + e) * S + 4 * q + l] = fma_(u32_to_unit(w[l]), span, lo);
}

// ------------------------------------------------------------------------------------------------ K4
// rank[i] = #{ j : f[j] > f[i]  or (f[j] == f[i] and j > i) }   -- O(n^2) counting, exact and
// order-independent (no sort).  Each offspring gets a 64-bit key (order-preserving image of the float in
// the high word, its index in the low word), so the whole tie rule is ONE unsigned 64-bit compare.
// 2-D grid: blockIdx.x picks 256 offspring i, blockIdx.y a slice of jt competitors j whose keys are
// wave-uniform and therefore fetched with scalar loads (no LDS, no vector memory in the loop); partial
// counts are combined with integer atomics (deterministic).
// History: v1 ran one block per 256 i over ALL j -- 16 workgroups on a 256-CU chip, 110 us at n = 4096;
// v2 staged float tiles in LDS -- LDS-issue bound, 674 us at n = 65 536.
__global__ __launch_bounds__(256) void k_rank_keys(const float *__restrict__ fit, int n,
                                                   unsigned long long *__restrict__ keys,
                                                   int32_t *__restrict__ rank)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const uint32_t u = f2u(fit[i] + 0.0f);                       // -0 -> +0 so that -0 == +0 stays a tie
    const uint32_t ordered = (u & 0x80000000u) ? ~u : (u | 0x80000000u);
    keys[i] = ((unsigned long long)ordered << 32) | (unsigned long long)(uint32_t)i;
    rank[i] = 0;
}

__global__ __launch_bounds__(256) void k_rank_count(const unsigned long long *__restrict__ keys, int n, int jt,
                                                    int32_t *__restrict__ rank)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    const int j0 = blockIdx.y * jt;
    const int lim = n - j0 < jt ? n - j0 : jt;
    const unsigned long long ki = keys[i < n ? i : n - 1];
    const unsigned long long *__restrict__ kj = keys + j0;       // uniform base: scalar loads below
    int count = 0;
    int k = 0;
    for (; k + 16 <= lim; k += 16) {
        unsigned long long c[16];
#pragma unroll
        for (int e = 0; e < 16; ++e) c[e] = kj[k + e];
#pragma unroll
        for (int e = 0; e < 16; ++e) count += (c[e] > ki) ? 1 : 0;
    }
    for (; k < lim; ++k) count += (kj[k] > ki) ? 1 : 0;
    if (i < n && count) atomicAdd(&rank[i], count);
}

// The same count with the keys formed on the fly from the fitness values (ses_openai_generation): the competitor's
// fitness is wave-uniform, so its key is built with scalar instructions next to the scalar load; rank[] must be zero on
// entry (the update kernel of the previous generation leaves it so).
__device__ __forceinline__ unsigned long long rank_key(uint32_t u, uint32_t index)
{
    u = u == 0x80000000u ? 0u : u;                                   // -0 -> +0 (what `fit + 0.0f` does in k_rank_keys)
    const uint32_t ordered = (u & 0x80000000u) ? ~u : (u | 0x80000000u);
    return ((unsigned long long)ordered << 32) | (unsigned long long)index;
}

__global__ __launch_bounds__(256) void k_rank_count_fitness(const float *__restrict__ fit, int n, int jt, int first,
                                                            int n_own, int32_t *__restrict__ rank)
{
    // rows [first, first + n_own) of the population are counted against all n competitors; rank[] is indexed from `first`
    // (first = 0, n_own = n: the whole population, every rank of a multi-GPU job the same; a shard's own rows only:
    // ses_openai_generation_sharded)
    const int il = blockIdx.x * 256 + threadIdx.x;
    const int i = first + il;
    const int j0 = blockIdx.y * jt;
    const int lim = n - j0 < jt ? n - j0 : jt;
    const int ic = i < n ? i : n - 1;
    const unsigned long long ki = rank_key(f2u(fit[ic]), (uint32_t)ic);
    const uint32_t *__restrict__ fj = reinterpret_cast<const uint32_t *>(fit) + j0;   // uniform base: scalar loads below
    int count = 0;
    int k = 0;
    for (; k + 16 <= lim; k += 16) {
        uint32_t c[16];
#pragma unroll
        for (int e = 0; e < 16; ++e) c[e] = fj[k + e];
#pragma unroll
        for (int e = 0; e < 16; ++e) count += (rank_key(c[e], (uint32_t)(j0 + k + e)) > ki) ? 1 : 0;
    }
    for (; k < lim; ++k) count += (rank_key(fj[k], (uint32_t)(j0 + k)) > ki) ? 1 : 0;
    if (il < n_own && count) atomicAdd(&rank[il], count);
}

// The same count with the episode mean folded in (ses_run_generations on one GPU, "fused_episode_mean"): no fitness vector
// exists yet -- the rollout left its per-episode returns, float64[n, E] -- so every workgroup first forms the 64-bit keys it
// needs from the means themselves, `jt` competitors and its own 256 rows, exactly as the episode-mean kernel forms them
// (sequential float64 sum, / E, rounded to float32: loop.py:124), stages the competitors' keys in LDS and counts from there
// (broadcast reads).  The workgroups of the first slice also write fitness[] (the gradient kernel and the caller read it) and
// the first thread the time stamp the mean kernel would have written.  One launch less per generation.
constexpr int RANK_EP_JT_MAX = 256;

__global__ __launch_bounds__(256) void k_rank_count_episodes(const double *__restrict__ ep_return, int E, int n, int jt,
                                                             int32_t *__restrict__ rank, float *__restrict__ fitness,
                                                             unsigned long long *__restrict__ stamp)
{
    __shared__ unsigned long long kj[RANK_EP_JT_MAX];
    if (stamp && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) *stamp = real_time();   // end of the rollout phase
    const int i = blockIdx.x * 256 + threadIdx.x;
    const int j0 = blockIdx.y * jt;
    const int lim = n - j0 < jt ? n - j0 : jt;
    auto mean_of = [&](int row) {
        double total = 0.0;
        for (int e = 0; e < E; ++e) total += ep_return[(size_t)row * E + e];
        return (float)(total / (double)E);
    };
    if ((int)threadIdx.x < lim) kj[threadIdx.x] = rank_key(f2u(mean_of(j0 + threadIdx.x)), (uint32_t)(j0 + threadIdx.x));
    const int ic = i < n ? i : n - 1;
    const float fi = mean_of(ic);
    if (blockIdx.y == 0 && i < n) fitness[i] = fi;
    const unsigned long long ki = rank_key(f2u(fi), (uint32_t)ic);
    __syncthreads();
    int count = 0;
    int k = 0;
    for (; k + 8 <= lim; k += 8) {
        unsigned long long c[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) c[e] = kj[k + e];
#pragma unroll
        for (int e = 0; e < 8; ++e) count += (c[e] > ki) ? 1 : 0;
    }
    for (; k < lim; ++k) count += (kj[k] > ki) ? 1 : 0;
    if (i < n && count) atomicAdd(&rank[i], count);
}

// The counting rank fed by the granules of a fused fitness exchange (ses_run_generations on several ranks, up to 8192 rows): no
// gathered vector exists -- every rank's episode-mean kernel has stored its values as granules into every rank's mailbox
// (k_fitness_mean_granules) -- so a workgroup polls the `jt` competitors of its slice and its own 256 rows there, stages the
// competitors' keys in LDS and counts.  The workgroups of the first row block also write fitness_all[] when the caller needs the
// vector (the replicated tail: its gradient kernel reads the fitness of the row of rank 0).
__global__ __launch_bounds__(256) void k_rank_count_granules(P2pGranuleView gv, int per_rank, int n, int jt, int first, int n_own,
                                                             int32_t *__restrict__ rank, float *__restrict__ fitness_all)
{
    __shared__ unsigned long long kj[RANK_EP_JT_MAX];
    const int il = blockIdx.x * 256 + threadIdx.x;
    const int i = first + il;
    const int j0 = blockIdx.y * jt;
    const int lim = n - j0 < jt ? n - j0 : jt;
    auto bits_of = [&](int row) {
        const int owner = row / per_rank;
        return granule_wait(gv.src + (size_t)owner * gv.section + (row - owner * per_rank), gv, owner);
    };
    if ((int)threadIdx.x < lim) {
        const uint32_t u = bits_of(j0 + threadIdx.x);
        kj[threadIdx.x] = rank_key(u, (uint32_t)(j0 + threadIdx.x));
        if (fitness_all && blockIdx.x == 0) fitness_all[j0 + threadIdx.x] = __builtin_bit_cast(float, u);
    }
    const int ic = i < n ? i : n - 1;
    const unsigned long long ki = rank_key(bits_of(ic), (uint32_t)ic);
    __syncthreads();
    int count = 0;
    int k = 0;
    for (; k + 8 <= lim; k += 8) {
        unsigned long long c[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) c[e] = kj[k + e];
#pragma unroll
        for (int e = 0; e < 8; ++e) count += (c[e] > ki) ? 1 : 0;
    }
    for (; k < lim; ++k) count += (kj[k] > ki) ? 1 : 0;
    if (il < n_own && count) atomicAdd(&rank[il], count);
}

// Large populations (n > RANK_SORT_MIN): sort tiles of RANK_TILE keys in LDS (bitonic network), then every
// offspring binary-searches each sorted tile for the number of larger keys.  O(n log^2 T + n (n/T) log T)
// instead of O(n^2).  Keys are distinct (index in the low
// word), padding keys are 0 and never count as larger.
constexpr int RANK_TILE = 1024;
constexpr int RANK_SORT_MIN = 8192;

// Bitonic sort of one tile of RANK_TILE = 1024 keys by 512 threads, two keys per thread, ascending.  Wave w owns the
// elements [128 w, 128 w + 128): lane l holds a = 128 w + l and b = a + 64, so every compare-exchange distance j <= 64 stays
// inside the wave -- j = 64 between the thread's own two keys, j < 64 through a lane exchange (ds_bpermute) -- and only
// the six stages with j >= 128 go through LDS, double-buffered so that each costs ONE barrier.  (First form: one
// compare-exchange per thread per stage in LDS, 55 stages x (two reads, two writes, a barrier): 9 us per launch.)
__device__ __forceinline__ void bitonic_keep(unsigned long long &x, unsigned long long y, bool keep_min)
{
    const bool x_gt = x > y;
    x = (x_gt == keep_min) ? y : x;
}

__device__ __forceinline__ void bitonic_sort_tile(unsigned long long &xa, unsigned long long &xb,
                                                  unsigned long long (*buf)[RANK_TILE])
{
    const int a = (threadIdx.x >> 6) * 128 + (threadIdx.x & 63), b = a + 64;
    int flip = 0;
#pragma unroll
    for (int k = 2; k <= RANK_TILE; k <<= 1) {
        const bool up_a = (a & k) == 0, up_b = (b & k) == 0;
#pragma unroll
        for (int j = k >> 1; j > 0; j >>= 1) {
            if (j >= 128) {
                buf[flip][a] = xa;
                buf[flip][b] = xb;
                __syncthreads();
                const unsigned long long ya = buf[flip][a ^ j], yb = buf[flip][b ^ j];
                flip ^= 1;
                bitonic_keep(xa, ya, ((a & j) == 0) == up_a);
                bitonic_keep(xb, yb, ((b & j) == 0) == up_b);
            } else if (j == 64) {
                // a < b = a | 64: ascending (up) keeps the smaller key in a
                const unsigned long long lo = xa < xb ? xa : xb, hi = xa < xb ? xb : xa;
                xa = up_a ? lo : hi;
                xb = up_a ? hi : lo;
            } else {
                const unsigned long long ya = __shfl_xor(xa, j, 64), yb = __shfl_xor(xb, j, 64);
                bitonic_keep(xa, ya, ((a & j) == 0) == up_a);
                bitonic_keep(xb, yb, ((b & j) == 0) == up_b);
            }
        }
    }
}

// keys formed from the fitness values on the fly (no key array, no key kernel); padding keys are 0 and never count as larger
__global__ __launch_bounds__(RANK_TILE / 2) void k_rank_tile_sort(const float *__restrict__ fit, int n,
                                                                  unsigned long long *__restrict__ sorted)
{
    __shared__ unsigned long long buf[2][RANK_TILE];
    const int base = blockIdx.x * RANK_TILE;
    const int a = (threadIdx.x >> 6) * 128 + (threadIdx.x & 63), b = a + 64;
    unsigned long long xa = base + a < n ? rank_key(f2u(fit[base + a]), (uint32_t)(base + a)) : 0ull;
    unsigned long long xb = base + b < n ? rank_key(f2u(fit[base + b]), (uint32_t)(base + b)) : 0ull;
    bitonic_sort_tile(xa, xb, buf);
    sorted[base + a] = xa;                                                          // ascending
    sorted[base + b] = xb;
}

// One workgroup = 256 offspring x one sorted tile: the tile (8 KB) is staged in LDS once and each thread binary-
// searches it there (10 dependent LDS reads instead of 10 dependent L2 reads: 36 -> see DESIGN.md at 32 768).
__global__ __launch_bounds__(256) void k_rank_search(const float *__restrict__ fit,
                                                     const unsigned long long *__restrict__ sorted, int n,
                                                     int32_t *__restrict__ rank)
{
    __shared__ unsigned long long t[RANK_TILE];
    const ulonglong2 *src = reinterpret_cast<const ulonglong2 *>(sorted + (size_t)blockIdx.y * RANK_TILE);
    for (int e = threadIdx.x; e < RANK_TILE / 2; e += 256) reinterpret_cast<ulonglong2 *>(t)[e] = src[e];
    __syncthreads();
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const unsigned long long ki = rank_key(f2u(fit[i]), (uint32_t)i);
    int lo = 0, hi = RANK_TILE;                      // first position with t[pos] > ki
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (t[mid] > ki) hi = mid; else lo = mid + 1;
    }
    if (lo < RANK_TILE) atomicAdd(&rank[i], RANK_TILE - lo);
}

// A shard's own rows against ALL tiles in one launch (ses_openai_generation_sharded): workgroup (x, y) sorts tile y
// itself -- the sort is repeated by the n_own / 1024 workgroups that share the tile, a few microseconds of otherwise idle
// CUs instead of a dependent launch -- and then searches it for its 1024 own rows, two per thread.
// GRAN: there is no gathered fitness vector.  The tile's values are polled as granules in this rank's mailbox, where the
// episode-mean kernel of the rank that owns the tile's rows has stored them (k_fitness_mean_granules; per_rank is a multiple
// of the tile, so a tile has one owner); `fit` then holds this rank's OWN values only, indexed from `first`.
template <bool GRAN>
__global__ __launch_bounds__(RANK_TILE / 2) void k_rank_sort_search(const float *__restrict__ fit, int n, int first,
                                                                    int n_own, int32_t *__restrict__ rank_own,
                                                                    P2pGranuleView gv = P2pGranuleView{}, int per_rank = 1)
{
    __shared__ unsigned long long buf[2][RANK_TILE];
    const int base = blockIdx.y * RANK_TILE;
    const int a = (threadIdx.x >> 6) * 128 + (threadIdx.x & 63), b = a + 64;
    unsigned long long xa = 0ull, xb = 0ull;
    if (GRAN) {
        const int owner = base / per_rank;
        const unsigned long long *src = gv.src + (size_t)owner * gv.section + (base - owner * per_rank);
        if (base + a < n) xa = rank_key(granule_wait(src + a, gv, owner), (uint32_t)(base + a));
        if (base + b < n) xb = rank_key(granule_wait(src + b, gv, owner), (uint32_t)(base + b));
    } else {
        xa = base + a < n ? rank_key(f2u(fit[base + a]), (uint32_t)(base + a)) : 0ull;
        xb = base + b < n ? rank_key(f2u(fit[base + b]), (uint32_t)(base + b)) : 0ull;
    }
    bitonic_sort_tile(xa, xb, buf);
    // the sort's sixth and last LDS stage used buf[1]; every thread has passed its barrier, so the fifth stage's reads of
    // buf[0] are over and buf[0] can take the sorted tile without another barrier
    buf[0][a] = xa;
    buf[0][b] = xb;
    __syncthreads();
    const unsigned long long *t = buf[0];
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        const int il = blockIdx.x * RANK_TILE + half * (RANK_TILE / 2) + threadIdx.x;
        if (il >= n_own) continue;
        const int i = first + il;
        const unsigned long long ki = rank_key(f2u(fit[GRAN ? il : i]), (uint32_t)i);
        int lo = 0, hi = RANK_TILE;
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (t[mid] > ki) hi = mid; else lo = mid + 1;
        }
        if (lo < RANK_TILE) atomicAdd(&rank_own[il], RANK_TILE - lo);
    }
}

__global__ __launch_bounds__(256) void k_rank_weights(const int32_t *__restrict__ rank, int n,
                                                      const float *__restrict__ fitness,
                                                      double *__restrict__ weights, float *__restrict__ best)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int r = rank[i];
    if (best && r == 0) *best = fitness[i];                           // max(rewards), loop.py:82-84 `best_reward`
    if (!weights) return;
    const double nm1 = (double)(n - 1);
    const double centred = ((double)(n - 1 - r) / nm1) - 0.5;         // offspring_strategies.py:394-396
    const double sd = sqrt((double)(n + 1) / (12.0 * nm1));           // closed-form std of the rank grid
    weights[i] = centred / sd;
}

// ------------------------------------------------------------------------------------------------ K5
// Adam exactly as optimizers.py:42-57 evaluates it under numpy >= 2 promotion rules:
// float32 moments, float64 step, float32 parameter store.
__device__ __forceinline__ void adam_apply(float g, double adam_a, float &mu, float &m, float &v)
{
    const float b1 = 0.99f, b2 = 0.999f;
    const float omb1 = (float)(1.0 - 0.99), omb2 = (float)(1.0 - 0.999);
    const float mn = (b1 * m) + (omb1 * g);
    const float vn = (b2 * v) + (omb2 * (g * g));
    const double num = (-adam_a) * (double)mn;
    const float den = __builtin_sqrtf(vn) + 1e-08f;
    const double step = num / (double)den;
    m = mn;
    v = vn;
    mu = (float)((double)mu + step);
}

// grad[p] = sum_i w_i * eps(i, p) with eps regenerated from Philox, in two stages:
//   stage 1: one 256-thread workgroup per (parameter quad, chunk of ES_CHUNK = 1024 offspring); thread c
//            accumulates rows c, c+256, ... of its chunk in ascending order, a fixed LDS tree combines the
//            256 partials -> partial[chunk][p]
//   stage 2: one thread per parameter adds the chunk partials in ascending chunk order, scales, applies Adam.
// The order depends only on n, never on the GPU count (every rank computes all n rows), so mu stays
// bit-identical across ranks.  (v1 was a single stage with one workgroup per quad: 57 workgroups, 141 us at
// n = 65 536.)
constexpr int ES_CHUNK = 1024;

__global__ __launch_bounds__(256) void k_es_grad_partial(const double *__restrict__ weights, int n, int skip_row0,
                                                         uint64_t seed, uint64_t gen, int P4,
                                                         float *__restrict__ partial)
{
    __shared__ float red[4][256];
    const int q = blockIdx.x;
    const int row0 = blockIdx.y * ES_CHUNK;
    const int row1 = row0 + ES_CHUNK < n ? row0 + ES_CHUNK : n;
    float acc[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    for (int i = row0 + threadIdx.x; i < row1; i += 256) {
        if (skip_row0 && i == 0) continue;
        const float w = (float)weights[i];
        float z[4];
        normal4(seed, gen, (uint32_t)i, (uint32_t)q, z);
#pragma unroll
        for (int l = 0; l < 4; ++l) acc[l] = fma_(w, z[l], acc[l]);
    }
#pragma unroll
    for (int l = 0; l < 4; ++l) red[l][threadIdx.x] = acc[l];
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (threadIdx.x < s) {
#pragma unroll
            for (int l = 0; l < 4; ++l) red[l][threadIdx.x] = red[l][threadIdx.x] + red[l][threadIdx.x + s];
        }
        __syncthreads();
    }
    if (threadIdx.x < 4) partial[(size_t)blockIdx.y * P4 + 4 * q + threadIdx.x] = red[threadIdx.x][0];
}

// (mu, m, v) may be updated in place (mu_out == mu ...) or into other buffers.
// One wavefront per parameter: lane c fetches the partial of chunk c (all chunks in flight at once), then the wave adds
// them in ascending chunk order through v_readlane -- the same sum, in the same order, as a thread walking the chunks,
// without its chain of dependent L2 reads (32 chunks: 9.9 -> ~3 us).
// The update over granules (ses_openai_generation_sharded on the peer-store transport): lane c WAITS for the granule of
// chunk c -- rank c / cl's section of this rank's mailbox, written by that rank's gradient kernel -- instead of loading a
// float from an all-gathered array; the ordered sum and Adam are k_es_apply's.  A chunk that does not arrive in time is NaN
// and marks its rank in the error words (the host's recovery is ESLoop.run's, as for the fitness exchange).
__global__ __launch_bounds__(256) void k_es_apply_granules(P2pGranuleView gv, int chunks, int P, int P4, float update_factor,
                                                           double adam_a, const float *mu, const float *m, const float *v,
                                                           float *mu_out, float *m_out, float *v_out, int cl,
                                                           float *__restrict__ best)
{
    const int p = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (p >= P) return;                                                  // wave-uniform
    if (p == 0) {                                                        // the candidates for max(fitness): exactly one is there
        for (int c = lane; c < gv.world * cl; c += 64) {
            const uint32_t u = granule_wait(gv.src + (size_t)(c / cl) * gv.section + (size_t)cl * P4 + c % cl, gv, c / cl);
            if (u != 0xFFFFFFFFu && best) *best = __builtin_bit_cast(float, u);
        }
    }
    float sum = 0.0f;
    for (int base = 0; base < chunks; base += 64) {
        const int cm = base + lane;
        float mine = 0.0f;
        if (cm < chunks)
            mine = __builtin_bit_cast(float, granule_wait(gv.src + (size_t)(cm / cl) * gv.section + (size_t)(cm % cl) * P4 + p, gv, cm / cl));
        const int cnt = chunks - base < 64 ? chunks - base : 64;
        for (int c = 0; c < cnt; ++c) {
            const float x = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, mine), c));
            sum = (base + c == 0) ? x : sum + x;
        }
    }
    if (lane != 0) return;
    const float g = sum * update_factor;  // offspring_strategies.py:414
    float muv = mu[p], mv = m[p], vv = v[p];
    adam_apply(g, adam_a, muv, mv, vv);
    mu_out[p] = muv; m_out[p] = mv; v_out[p] = vv;
}

__global__ __launch_bounds__(256) void k_es_apply(const float *__restrict__ partial, int chunks, int P, int P4,
                                                  float update_factor, double adam_a, const float *mu, const float *m,
                                                  const float *v, float *mu_out, float *m_out, float *v_out,
                                                  float *__restrict__ grad_out, int cl, int stride, int n_cand,
                                                  float *__restrict__ best)
{
    // chunk c's partial sits at partial[(c / cl) * stride + (c % cl) * P4 + p]: the all-gathered payloads of the ranks
    // (cl chunks each, `stride` floats apart: ses_openai_generation_sharded) -- or one contiguous [chunks, P4] array
    // (cl = chunks).  n_cand > 0: behind each rank's cl * P4 partials lie cl best-reward candidates as bit patterns
    // (0xFFFFFFFF = none); the one that is there is max(fitness).
    const int p = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (p >= P) return;                                                  // wave-uniform
    if (n_cand > 0 && p == 0) {
        for (int c = lane; c < n_cand; c += 64) {
            const uint32_t u = reinterpret_cast<const uint32_t *>(partial)[(size_t)(c / cl) * stride + (size_t)cl * P4 + c % cl];
            if (u != 0xFFFFFFFFu && best) *best = __builtin_bit_cast(float, u);
        }
    }
    float sum = 0.0f;
    for (int base = 0; base < chunks; base += 64) {
        const int cm = base + lane;
        const float mine = cm < chunks ? partial[(size_t)(cm / cl) * stride + (size_t)(cm % cl) * P4 + p] : 0.0f;
        const int cnt = chunks - base < 64 ? chunks - base : 64;
        for (int c = 0; c < cnt; ++c) {
            const float x = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, mine), c));
            sum = (base + c == 0) ? x : sum + x;
        }
    }
    if (lane != 0) return;
    const float g = sum * update_factor;  // offspring_strategies.py:414
    if (grad_out) grad_out[p] = g;
    float muv = mu[p], mv = m[p], vv = v[p];
    adam_apply(g, adam_a, muv, mv, vv);
    mu_out[p] = muv; m_out[p] = mv; v_out[p] = vv;
}

// ---- the openai_es fitness loop in four launches (ses_openai_generation) -----------------------------------------
// Same arithmetic as k_rank_weights + k_es_grad_partial + k_es_apply + k_perturb, regrouped so that a generation needs
// four small launches after the rollout instead of seven (six above 8192 rows: sort + search rank):
//   * the rank-centring weight of a row is a closed form of its rank: formed where it is used (stage 1 of the gradient)
//     instead of being written and re-read; the thread that meets rank 0 also reports best = max(fitness);
//   * Adam's update of (mu, m, v) -- P parameters -- is k_es_apply, one wavefront per parameter (chunk partials added in
//     ascending order); old and new vectors are distinct buffers (the caller ping-pongs).  FINAL == true moves it into
//     this kernel instead (tuning knob "es_final_max_chunks"; measured slower, kept as a tested variant).  (An earlier
//     form recomputed the update in every thread of the perturbation kernel: that kernel then took as long as the two it
//     replaced, and O(n x chunks) reads made it 1 ms at 65 536 offspring.)
//   * the keys of the counting rank are formed inside the count from the fitness values, the rank vector is cleared for
//     the next generation by the perturbation kernel: no key kernel, no memset.
// final == true: of the `chunks` workgroups that share a parameter quad, the one that finishes last (a ticket from the
// quad's atomic counter) adds the chunk partials in ascending order and applies Adam to those 4 parameters -- (mu, m,
// v)_in -> (mu, m, v)_out -- so that the launch that follows only has to perturb the new mu (the "last block done"
// reduction: writers fence before taking their ticket, the finisher fences before it reads and re-arms the counter).
// (Measured: 11.9 us per launch against 7.4 for the plain partial-sum kernel -- the agent-scope fences flush the XCD's L2
// -- i.e. what the separate update launch it replaces cost.)
template <bool FINAL, bool GRAN = false>
__global__ __launch_bounds__(256) void k_es_grad_partial_ranked(const int32_t *__restrict__ rank,
                                                                const float *__restrict__ fitness, int n, int skip_row0,
                                                                uint64_t seed, uint64_t gen, int P4,
                                                                float *partial, float *__restrict__ best,
                                                                unsigned int *counter, int chunks, int P,
                                                                float update_factor, double adam_a, const float *mu_in,
                                                                const float *m_in, const float *v_in, float *mu_out,
                                                                float *m_out, float *v_out, int first, int row_end,
                                                                uint32_t *__restrict__ cand_out, P2pGranuleView gv = P2pGranuleView{},
                                                                int cl = 0)
{
    // GRAN: the chunk partials (and the candidate) do not go to partial[] / cand_out[] but, as {sequence, value} granules,
    // straight into the mailbox of EVERY rank (peer stores over xGMI; this rank's own mailbox included): granule
    // blockIdx.y * P4 + 4 q + t of this rank's section, the candidates behind the cl * P4 partials.  The update kernel of each
    // rank polls them where they land: no exchange launch between the two kernels.
    // Shard form (ses_openai_generation_sharded): blockIdx.y counts the chunks of THIS rank's rows [first, row_end) --
    // first is a multiple of ES_CHUNK, so they are chunks of the global population --, rank[] is indexed from `first`, and
    // the chunk that holds the row of rank 0 reports its fitness as a bit pattern in cand_out[chunk] (0xFFFFFFFF = "not
    // here"): the rows of the other ranks are not seen, the candidates travel with the partials.  Replicated form: first =
    // 0, row_end = n, cand_out = null.
    __shared__ float red[4][256];
    __shared__ unsigned int ticket;
    __shared__ uint32_t cand;
    const int q = blockIdx.x;
    const int row0 = first + blockIdx.y * ES_CHUNK;
    const int row1 = row0 + ES_CHUNK < row_end ? row0 + ES_CHUNK : row_end;
    const double nm1 = (double)(n - 1);
    const double sd = sqrt((double)(n + 1) / (12.0 * nm1));           // closed-form std of the rank grid
    if ((cand_out || GRAN) && q == 0) {                                // uniform per workgroup
        if (threadIdx.x == 0) cand = 0xFFFFFFFFu;
        __syncthreads();
    }
    float acc[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    for (int i = row0 + threadIdx.x; i < row1; i += 256) {
        const int r = rank[i - first];
        if (q == 0 && r == 0) {                                        // max(rewards), loop.py:82-84 `best_reward`
            if (best) *best = fitness[i];
            if (cand_out || GRAN) cand = f2u(fitness[i]);
        }
        if (skip_row0 && i == 0) continue;
        const double centred = ((double)(n - 1 - r) / nm1) - 0.5;      // offspring_strategies.py:394-396
        const float w = (float)(centred / sd);
        float z[4];
        normal4(seed, gen, (uint32_t)i, (uint32_t)q, z);
#pragma unroll
        for (int l = 0; l < 4; ++l) acc[l] = fma_(w, z[l], acc[l]);
    }
#pragma unroll
    for (int l = 0; l < 4; ++l) red[l][threadIdx.x] = acc[l];
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (threadIdx.x < s) {
#pragma unroll
            for (int l = 0; l < 4; ++l) red[l][threadIdx.x] = red[l][threadIdx.x] + red[l][threadIdx.x + s];
        }
        __syncthreads();
    }
    if (GRAN) {
        const int peer = threadIdx.x >> 2, comp = threadIdx.x & 3;
        if (peer < gv.world) {                                         // thread (peer, component): one 8-byte store each
            granule_store(gv.dst[peer] + (size_t)blockIdx.y * P4 + 4 * q + comp, gv.seq, f2u(red[comp][0]));
            if (q == 0 && comp == 0) granule_store(gv.dst[peer] + (size_t)cl * P4 + blockIdx.y, gv.seq, cand);
        }
        return;
    }
    if (threadIdx.x < 4) partial[(size_t)blockIdx.y * P4 + 4 * q + threadIdx.x] = red[threadIdx.x][0];
    if (cand_out && q == 0 && threadIdx.x == 0) cand_out[blockIdx.y] = cand;
    if (!FINAL) return;
    // ---- last workgroup of this quad done: finish the update of its 4 parameters ----
    if (threadIdx.x < 4) __threadfence();                              // this workgroup's partials are visible device-wide
    __syncthreads();
    if (threadIdx.x == 0) ticket = atomicAdd(counter + q, 1u);
    __syncthreads();
    if (ticket != gridDim.y - 1 || threadIdx.x >= 4) return;
    __threadfence();                                                   // ... before anybody else's are read
    if (threadIdx.x == 0) counter[q] = 0u;                             // armed for the next generation
    const int p = 4 * q + threadIdx.x;
    if (p >= P) return;
    float sum = __hip_atomic_load(partial + p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // device-coherent read
    for (int c = 1; c < chunks; ++c)
        sum = sum + __hip_atomic_load(partial + (size_t)c * P4 + p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const float g = sum * update_factor;                               // offspring_strategies.py:414
    float muv = mu_in[p], mv = m_in[p], vv = v_in[p];
    adam_apply(g, adam_a, muv, mv, vv);
    mu_out[p] = muv; m_out[p] = mv; v_out[p] = vv;
}


// k_perturb for the openai_es population shape: global row 0 = mu, every other row mu + sigma * eps
__global__ __launch_bounds__(256) void k_perturb_openai(const float *__restrict__ mu, float sigma, uint64_t seed,
                                                        uint64_t gen, long long first_row, int n_rows, int P, int quads,
                                                        float *__restrict__ theta, unsigned long long *__restrict__ stamp,
                                                        int32_t *__restrict__ rank_to_clear, int n_clear)
{
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (stamp && t == 0) *stamp = real_time();
    // the rank vector has been consumed by the gradient kernel: leave it zeroed for the next generation's count
    for (long long i = t; i < n_clear; i += (long long)gridDim.x * blockDim.x) rank_to_clear[i] = 0;
    if (t >= (long long)n_rows * quads) return;
    const int i = (int)(t / quads);
    const int q = (int)(t - (long long)i * quads);
    const int lim = P - 4 * q < 4 ? P - 4 * q : 4;
    float *dst = theta + (size_t)i * P + 4 * q;
    const long long row = first_row + i;
    if (row == 0) {
        for (int l = 0; l < lim; ++l) dst[l] = mu[4 * q + l];
        return;
    }
    float z[4];
    normal4(seed, gen, (uint32_t)row, (uint32_t)q, z);
    for (int l = 0; l < lim; ++l) dst[l] = fma_(sigma, z[l], mu[4 * q + l]);
}

// k_es_apply and k_perturb_openai in one launch, for policies of up to APPLY_PERTURB_MAX_P parameters: EVERY workgroup forms
// the whole new mean itself -- a thread per parameter: the chunk partials added in ascending order, the factor, Adam, exactly
// k_es_apply's arithmetic -- into LDS and perturbs from there; workgroup 0 also stores mu / m / v.  The update is ~100
// instructions per parameter on values that sit in L2, repeated by ~900 workgroups: cheaper than the launch it replaces
// (4.7 us per generation of the headline).  The in and out vectors are distinct buffers (ses_openai_generation requires it),
// so no workgroup can read a value another one has already replaced.
constexpr int APPLY_PERTURB_MAX_P = 1024;
constexpr int APPLY_PERTURB_MAX_CHUNKS = 16;
__global__ __launch_bounds__(256) void k_es_apply_perturb(const float *__restrict__ partial, int chunks, int P4, float update_factor,
                                                          double adam_a, const float *__restrict__ mu_in,
                                                          const float *__restrict__ m_in, const float *__restrict__ v_in,
                                                          float *__restrict__ mu_out, float *__restrict__ m_out,
                                                          float *__restrict__ v_out, float sigma, uint64_t seed, uint64_t gen,
                                                          long long first_row, int n_rows, int P, int quads,
                                                          float *__restrict__ theta, unsigned long long *__restrict__ stamp,
                                                          int32_t *__restrict__ rank_to_clear, int n_clear)
{
    __shared__ float mu_new[APPLY_PERTURB_MAX_P];
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (stamp && t == 0) *stamp = real_time();
    for (int p = threadIdx.x; p < P; p += 256) {
        float sum = partial[p];
        for (int c = 1; c < chunks; ++c) sum = sum + partial[(size_t)c * P4 + p];
        const float g = sum * update_factor;                              // offspring_strategies.py:414
        float muv = mu_in[p], mv = m_in[p], vv = v_in[p];
        adam_apply(g, adam_a, muv, mv, vv);
        mu_new[p] = muv;
        if (blockIdx.x == 0) { mu_out[p] = muv; m_out[p] = mv; v_out[p] = vv; }
    }
    for (long long i = t; i < n_clear; i += (long long)gridDim.x * blockDim.x) rank_to_clear[i] = 0;
    __syncthreads();
    if (t >= (long long)n_rows * quads) return;
    const int i = (int)(t / quads);
    const int q = (int)(t - (long long)i * quads);
    const int lim = P - 4 * q < 4 ? P - 4 * q : 4;
    float *dst = theta + (size_t)i * P + 4 * q;
    const long long row = first_row + i;
    if (row == 0) {
        for (int l = 0; l < lim; ++l) dst[l] = mu_new[4 * q + l];
        return;
    }
    float z[4];
    normal4(seed, gen, (uint32_t)row, (uint32_t)q, z);
    for (int l = 0; l < lim; ++l) dst[l] = fma_(sigma, z[l], mu_new[4 * q + l]);
}

// Reference-order accumulation over stored (mu + eps) rows: one thread per parameter, sequential over
// offspring, float32 accumulator with float64 products (offspring_strategies.py:409-414 under numpy 2).
__global__ __launch_bounds__(64) void k_es_update_stored(const double *__restrict__ weights, int n,
                                                         const float *__restrict__ eps_store, int P,
                                                         float update_factor, double adam_a, float *__restrict__ mu,
                                                         float *__restrict__ m, float *__restrict__ v,
                                                         float *__restrict__ grad_out)
{
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= P) return;
    float acc = 0.0f;
    for (int i = 0; i < n; ++i) {
        const double prod = (double)eps_store[(size_t)i * P + p] * weights[i];
        acc = (float)((double)acc + prod);
    }
    const float g = acc * update_factor;
    if (grad_out) grad_out[p] = g;
    float muv = mu[p], mv = m[p], vv = v[p];
    adam_apply(g, adam_a, muv, mv, vv);
    mu[p] = muv; m[p] = mv; v[p] = vv;
}

// ------------------------------------------------------------------------------------------------ K6
__global__ void k_elite_ids(const int32_t *__restrict__ rank, int n, int k, int32_t *__restrict__ elite_ids)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n && rank[i] < k) elite_ids[rank[i]] = i;
}

// one workgroup: the ids are needed together (the alias flags compare every elite with elite 0)
__global__ __launch_bounds__(1024) void k_elite_select(const int32_t *__restrict__ rank, int n, int k,
                                                       const int32_t *__restrict__ parent_map,
                                                       int32_t *__restrict__ alias_state, int32_t *__restrict__ elite_ids,
                                                       int32_t *__restrict__ elite_parent_idx,
                                                       int32_t *__restrict__ alias_first)
{
    __shared__ int32_t ids[1024];
    for (int i = threadIdx.x; i < n; i += 1024) {
        const int r = rank[i];
        if (r < k) ids[r] = i;
    }
    __syncthreads();
    const int j = threadIdx.x;
    const int first = ids[0];
    const int state = alias_state ? *alias_state : 0;
    if (j < k) {
        const int id = ids[j];
        elite_ids[j] = id;
        elite_parent_idx[j] = parent_map[id];
        if (alias_first)
            alias_first[j] = (j > 0 && state != 0 && (first == 0 || first == 1) && (id == 0 || id == 1) && id != first) ? 1 : 0;
    }
    __syncthreads();                                            // every thread has read the old state
    if (j == 0 && alias_state) *alias_state = (first == 0 || (first == 1 && state != 0)) ? 1 : 0;
}

// ---- the elite strategies' tail for small populations in two launches (ses_run_generations on one GPU, n <= 1024) -----------------
// One workgroup does what was five launches: the episode mean of every row (the rollout left its per-episode returns), the
// counting rank with all keys in LDS, max(fitness), and ses_elite_select's bookkeeping (elite ids, their parent-map entries, the
// aliasing flags and state of simple_evolution).  Same arithmetic and tie rule as k_fitness_mean + k_rank_keys + k_rank_count +
// k_rank_weights + k_elite_select.  The reference's configs have 96-257 rows: a generation is latency, not work.
__global__ __launch_bounds__(1024) void k_elite_rank_select_small(const double *__restrict__ ep_return, int E, int n, int k,
                                                                  const int32_t *__restrict__ parent_map,
                                                                  int32_t *__restrict__ alias_state, int32_t *__restrict__ rank,
                                                                  float *__restrict__ fitness, float *__restrict__ best,
                                                                  int32_t *__restrict__ elite_ids,
                                                                  int32_t *__restrict__ elite_parent_idx,
                                                                  int32_t *__restrict__ alias_first,
                                                                  unsigned long long *__restrict__ stamp)
{
    __shared__ unsigned long long keys[1024];
    __shared__ int32_t ids[1024];
    const int i = threadIdx.x;
    if (stamp && i == 0) *stamp = real_time();                           // end of the rollout phase
    float fi = 0.0f;
    if (i < n) {
        double total = 0.0;
        for (int e = 0; e < E; ++e) total += ep_return[(size_t)i * E + e];
        fi = (float)(total / (double)E);                                 // loop.py:124
        fitness[i] = fi;
        keys[i] = rank_key(f2u(fi), (uint32_t)i);
    }
    __syncthreads();
    if (i < n) {
        const unsigned long long ki = keys[i];
        int r = 0;
        for (int j = 0; j < n; ++j) r += (keys[j] > ki) ? 1 : 0;
        rank[i] = r;
        if (r == 0 && best) *best = fi;
        if (r < k) ids[r] = i;
    }
    __syncthreads();
    const int first = ids[0];
    const int state = alias_state ? *alias_state : 0;
    if (i < k) {
        const int id = ids[i];
        elite_ids[i] = id;
        elite_parent_idx[i] = parent_map[id];
        if (alias_first)
            alias_first[i] = (i > 0 && state != 0 && (first == 0 || first == 1) && (id == 0 || id == 1) && id != first) ? 1 : 0;
    }
    __syncthreads();                                            // every thread has read the old state
    if (i == 0 && alias_state) *alias_state = (first == 0 || (first == 1 && state != 0)) ? 1 : 0;
}

// simple_evolution's new parent without materialising the elite rows: mean[p] = ((row_0 + row_1) + ...) / k with row_j =
// parents[idx_j] + sigma * eps(seed, gen, ids_j, p) -- k_perturb's fma and k_elite_mean's in-place order and aliasing rule
// (offspring_strategies.py:234-248), one thread per parameter quad, the elites' noise regenerated where it is summed.
__global__ __launch_bounds__(64) void k_elite_mean_philox(const float *__restrict__ parents, const int32_t *__restrict__ parent_idx,
                                                          const int32_t *__restrict__ row_ids, const int32_t *__restrict__ alias_first,
                                                          int k, float sigma, uint64_t seed, uint64_t gen, int P, int quads,
                                                          float *__restrict__ mean)
{
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= quads) return;
    const int lim = P - 4 * q < 4 ? P - 4 * q : 4;
    float acc[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    for (int j = 0; j < k; ++j) {
        const int32_t pi = parent_idx[j];
        const float *src = parents + (size_t)(pi >= 0 ? pi : -1 - pi) * P + 4 * q;
        float row[4] = {0.0f, 0.0f, 0.0f, 0.0f};
        if (pi < 0) {
            for (int l = 0; l < lim; ++l) row[l] = src[l];
        } else {
            float z[4];
            normal4(seed, gen, (uint32_t)row_ids[j], (uint32_t)q, z);
            for (int l = 0; l < lim; ++l) row[l] = fma_(sigma, z[l], src[l]);
        }
        const bool alias = j > 0 && alias_first && alias_first[j];
#pragma unroll
        for (int l = 0; l < 4; ++l) {
            const float other = alias ? acc[l] : row[l];
            acc[l] = j == 0 ? row[l] : acc[l] + other;          // offspring_strategies.py:245  mu_param += elite_param
        }
    }
    for (int l = 0; l < lim; ++l) mean[4 * q + l] = acc[l] / (float)k;   // :248  param /= self.elite_num
}

__global__ void k_elite_mean(const float *__restrict__ rows, const int32_t *__restrict__ alias_first, int k, int P,
                             float *__restrict__ mean)
{
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= P) return;
    float acc = rows[p];
    for (int j = 1; j < k; ++j) {
        const float other = (alias_first && alias_first[j]) ? acc : rows[(size_t)j * P + p];
        acc = acc + other;                       // offspring_strategies.py:245  mu_param += elite_param
    }
    mean[p] = acc / (float)k;                    // :248  param /= self.elite_num
}

__global__ void k_gather_rows(const float *__restrict__ src, const int32_t *__restrict__ ids, int n_ids, int P,
                              float *__restrict__ dst)
{
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (long long)n_ids * P) return;
    const int i = (int)(t / P);
    const int p = (int)(t - (long long)i * P);
    dst[t] = src[(size_t)ids[i] * P + p];
}

}  // namespace ses

// ses_run_generations, elite strategies on one GPU with at most 1024 rows: [episode mean + rank + best + elite selection] in one
// launch and, for simple_evolution, [elite rows + their in-place mean] in a second (k_elite_rank_select_small,
// k_elite_mean_philox).  rows_out != null (simple_genetic): the elite rows are materialised by ses_perturb as before.
namespace ses {
int elite_tail_small(ses_handle *h, const double *ep_return, int32_t n, int32_t k, const int32_t *parent_map, int32_t *alias_state,
                     int32_t *rank, float *fitness, float *best, int32_t *ids, int32_t *pidx, int32_t *alias,
                     unsigned long long *stamp, const float *parents, float sigma, uint64_t seed, uint64_t gen, float *mean_out)
{
    SES_REQUIRE(n >= 2 && n <= 1024 && k >= 1 && k <= n, "elite_tail_small: 2 <= n <= 1024, 1 <= k <= n");
    SES_HIP_TRY(hipSetDevice(h->cfg.device));
    hipLaunchKernelGGL(k_elite_rank_select_small, dim3(1), dim3(1024), 0, h->stream, ep_return, h->cfg.eval_ep_num, n, k, parent_map,
                       alias_state, rank, fitness, best, ids, pidx, alias, stamp);
    if (mean_out) {
        const int quads = (h->P + 3) / 4;
        hipLaunchKernelGGL(k_elite_mean_philox, dim3(ceil_div(quads, 64)), dim3(64), 0, h->stream, parents, pidx, ids, alias, k, sigma,
                           seed, gen, h->P, quads, mean_out);
    }
    SES_HIP_TRY(hipGetLastError());
    return SES_OK;
}
}  // namespace ses

// ses_run_generations: may the fitness exchange of this layout be fused into the episode-mean kernel (producer) and
// k_rank_sort_search (consumer)?  Needs the sort path of the shard form, and a consumer grid that stays a fraction of the chip:
// its workgroups spin until the peers' values are there, and ranks that SHARE a GPU (the test rigs) must leave room for the
// rollouts they wait for.
namespace ses {
int openai_fused_fitness_ok(const ses_handle *h, int32_t n, int32_t per_rank, int32_t n_ranked)
{
    // n_ranked: the rows this rank ranks (its own in the shard form of the tail, all n in the replicated one)
    if (!h->tune_fused_fitness) return 0;
    if (n <= RANK_SORT_MIN)                                             // counting rank: k_rank_count_granules
        return (long long)ceil_div(n_ranked, 256) * ceil_div(n, RANK_EP_JT_MAX) <= 512 ? 1 : 0;
    if (n_ranked == n || per_rank % RANK_TILE != 0) return 0;           // sort path: the shard form only (k_rank_sort_search<true>)
    return (long long)ceil_div(per_rank, RANK_TILE) * ceil_div(n, RANK_TILE) <= 512 ? 1 : 0;
}
}  // namespace ses

extern "C" {

using namespace ses;

int ses_perturb(ses_handle *h, const float *parents, const int32_t *parent_idx, const int32_t *row_ids, float sigma,
                uint64_t seed, uint64_t gen, int64_t first_row, int32_t n_rows, float *theta)
{
    SES_REQUIRE(h && parents && theta, "ses_perturb: null argument");
    SES_REQUIRE(n_rows >= 1 && first_row >= 0 && first_row + n_rows <= (1ll << 30), "ses_perturb: row range");
    SES_HIP_TRY(hipSetDevice(h->cfg.device));
    const int quads = (h->P + 3) / 4;
    const long long threads = (long long)n_rows * quads;
    hipLaunchKernelGGL(k_perturb, dim3(ceil_div(threads, 256)), dim3(256), 0, h->stream, parents, parent_idx, row_ids,
                       sigma, seed, gen, (long long)first_row, n_rows, h->P, quads, theta, h->stamp);
    SES_HIP_TRY(hipGetLastError());
    return SES_OK;
}

int ses_noise(ses_handle *h, uint64_t seed, uint64_t gen, int64_t first_row, int32_t n_rows, float *eps)
{
    SES_REQUIRE(h && eps, "ses_noise: null argument");
    SES_REQUIRE(n_rows >= 1 && first_row >= 0 && first_row + n_rows <= (1ll << 30), "ses_noise: row range");
    SES_HIP_TRY(hipSetDevice(h->cfg.device));
    const int quads = (h->P + 3) / 4;
    const long long threads = (long long)n_rows * quads;
    hipLaunchKernelGGL(k_noise, dim3(ceil_div(threads, 256)), dim3(256), 0, h->stream, seed, gen, (long long)first_row,
                       n_rows, h->P, quads, eps);
    SES_HIP_TRY(hipGetLastError());
    return SES_OK;
}

int ses_perturb_host_noise(ses_handle *h, const float *parents, const int32_t *parent_idx, const double *eps64,
                           double sigma, int32_t n_rows, float *theta, float *eps_store)
{
    SES_REQUIRE(h && parents && eps64 && theta, "ses_perturb_host_noise: null argument");
    SES_REQUIRE(n_rows >= 1, "ses_perturb_host_noise: n_rows must be >= 1");
    SES_HIP_TRY(hipSetDevice(h->cfg.device));
    const long long threads = (long long)n_rows * h->P;
    hipLaunchKernelGGL(k_perturb_host_noise, dim3(ceil_div(threads, 256)), dim3(256), 0, h->stream, parents,
                       parent_idx, eps64, sigma, n_rows, h->P, theta, eps_store, h->stamp);
    SES_HIP_TRY(hipGetLastError());
    return SES_OK;
}

int ses_init_states_uniform(ses_handle *h, uint64_t seed, uint64_t gen, int64_t first_row, int32_t n_rows,
                            int32_t shared, int32_t width, float lo, float hi, float *out)
{
    SES_REQUIRE(h && out, "ses_init_states_uniform: null argument");
    SES_REQUIRE(n_rows >= 1 && first_row >= 0, "ses_init_states_uniform: row range");
    SES_HIP_TRY(hipSetDevice(h->cfg.device));
    const int S = width, E = h->cfg.eval_ep_num;
    SES_REQUIRE(E * 8 < (1 << 30) && S >= 1 && S <= 32, "ses_init_states_uniform: shape");
    const long long threads = (long long)n_rows * E * ((S + 3) / 4);
    hipLaunchKernelGGL(k_init_states_uniform, dim3(ceil_div(threads, 256)), dim3(256), 0, h->stream, seed, gen,
                       (long long)first_row, n_rows, E, S, shared, lo, hi - lo, out);
    SES_HIP_TRY(hipGetLastError());
    return SES_OK;
}

int ses_init_states_uniform_gens(ses_handle *h, uint64_t seed, uint64_t gen0, int32_t gens, int64_t first_row, int32_t n_rows,
                                 int32_t shared, int32_t width, float lo, float hi, float *out)
{
    SES_REQUIRE(h && out, "ses_init_states_uniform_gens: null argument");
    SES_REQUIRE(gens >= 1 && gens <= 65535 && n_rows >= 1 && first_row >= 0, "ses_init_states_uniform_gens: bad range");
    SES_HIP_TRY(hipSetDevice(h->cfg.device));
    const int S = width, E = h->cfg.eval_ep_num;
    SES_REQUIRE(E * 8 < (1 << 30) && S >= 1 && S <= 32, "ses_init_states_uniform_gens: shape");
    const long long threads = (long long)n_rows * E * ((S + 3) / 4);
    hipLaunchKernelGGL(k_init_states_uniform, dim3(ceil_div(threads, 256), gens), dim3(256), 0, h->stream, seed, gen0,
                       (long long)first_row, n_rows, E, S, shared, lo, hi - lo, out);
    SES_HIP_TRY(hipGetLastError());
    return SES_OK;
}

int ses_rank_center(ses_handle *h, const float *fitness, int32_t n, int32_t *rank, double *weights, float *best)
{
    SES_REQUIRE(h && fitness && rank, "ses_rank_center: null argument");
    SES_REQUIRE(n >= 2, "ses_rank_center: need at least 2 offspring (the reference divides by n-1)");
    SES_HIP_TRY(hipSetDevice(h->cfg.device));
    // ~2048 workgroups: j-slice length jt = n^2 / (256 * 2048), at least 64, multiple of 64
    long long jt = ((long long)n * n / (256ll * 2048ll) + 63) / 64 * 64;
    if (jt < 64) jt = 64;
    if (jt > 8192) jt = 8192;
    const int tiles = ceil_div(n, RANK_TILE);
    const size_t key_bytes = (sizeof(unsigned long long) * (size_t)n + 255) / 256 * 256;
    const int rc = ensure_reduce_scratch(h, key_bytes + sizeof(unsigned long long) * (size_t)tiles * RANK_TILE);
    if (rc != SES_OK) return rc;
    h->rank_zeroed = nullptr;                       // this call lays the scratch out differently from ses_openai_generation
    h->counter_armed = nullptr;
    if (n > RANK_SORT_MIN) {
        // sort + search, both with the keys formed from the fitness values on the fly
        unsigned long long *sorted = (unsigned long long *)((char *)h->red_scratch + key_bytes);
        SES_HIP_TRY(hipMemsetAsync(rank, 0, sizeof(int32_t) * (size_t)n, h->stream));
        hipLaunchKernelGGL(k_rank_tile_sort, dim3(tiles), dim3(RANK_TILE / 2), 0, h->stream, fitness, n, sorted);
        hipLaunchKernelGGL(k_rank_search, dim3(ceil_div(n, 256), tiles), dim3(256), 0, h->stream, fitness, sorted, n, rank);
    } else {
        unsigned long long *keys = (unsigned long long *)h->red_scratch;
        hipLaunchKernelGGL(k_rank_keys, dim3(ceil_div(n, 256)), dim3(256), 0, h->stream, fitness, n, keys, rank);
        hipLaunchKernelGGL(k_rank_count, dim3(ceil_div(n, 256), ceil_div(n, jt)), dim3(256), 0, h->stream, keys, n,
                           (int)jt, rank);
    }
    if (weights || best)
        hipLaunchKernelGGL(k_rank_weights, dim3(ceil_div(n, 256)), dim3(256), 0, h->stream, rank, n, fitness, weights,
                           best);
    SES_HIP_TRY(hipGetLastError());
    return SES_OK;
}

int ses_es_update_philox(ses_handle *h, const double *weights, int32_t n, int32_t skip_row0, uint64_t seed,
                         uint64_t gen, double lr, double sigma, double adam_a, float *mu, float *m, float *v,
                         float *grad_out)
{
    SES_REQUIRE(h && weights && mu && m && v, "ses_es_update_philox: null argument");
    SES_REQUIRE(n >= 1 && sigma != 0.0, "ses_es_update_philox: bad n / sigma");
    SES_HIP_TRY(hipSetDevice(h->cfg.device));
    // offspring_strategies.py:406-408: python-float factor, applied to a float32 array (weak scalar -> f32)
    double uf = lr / ((double)n * sigma);
    uf *= -1.0;
    const int quads = (h->P + 3) / 4, P4 = 4 * quads;
    const int chunks = ceil_div(n, ES_CHUNK);
    // partial sums live behind the rank keys in the handle's scratch (keys: n * 8 bytes)
    const size_t key_bytes = (sizeof(unsigned long long) * (size_t)n + 255) / 256 * 256;
    const int rc = ensure_reduce_scratch(h, key_bytes + sizeof(float) * (size_t)chunks * P4);
    if (rc != SES_OK) return rc;
    h->rank_zeroed = nullptr;
    h->counter_armed = nullptr;
    float *partial = (float *)((char *)h->red_scratch + key_bytes);
    hipLaunchKernelGGL(k_es_grad_partial, dim3(quads, chunks), dim3(256), 0, h->stream, weights, n, skip_row0, seed, gen,
                       P4, partial);
    hipLaunchKernelGGL(k_es_apply, dim3(ceil_div(h->P, 4)), dim3(256), 0, h->stream, partial, chunks, h->P, P4,
                       (float)uf, adam_a, mu, m, v, mu, m, v, grad_out, chunks, 0, 0, (float *)nullptr);
    SES_HIP_TRY(hipGetLastError());
    return SES_OK;
}

// The openai_es tail.  comm == null: replicated (every rank ranks all n rows and regenerates all n x P normals).
// comm != null: the shard form -- this rank ranks and accumulates only its own rows [first_row, first_row + n_rows), whose
// per_rank-row slot is a whole number of the gradient's 1024-row chunks; the ranks all-gather their chunk partials (+ the
// best-reward candidates) over `comm`, and the unchanged ordered update adds the chunks in ascending order: bit-identical
// to the replicated form for any world size.
static int openai_generation_impl(ses_handle *h, ses_handle *comm, const float *fitness, int32_t n, uint64_t seed, uint64_t gen,
                                  double lr, double sigma, double adam_a, const float *mu_in, const float *m_in,
                                  const float *v_in, float *mu_out, float *m_out, float *v_out, float next_sigma,
                                  uint64_t next_gen, int64_t first_row, int32_t n_rows, int32_t per_rank, int32_t world,
                                  float *theta_next, float *best)
{
    SES_HIP_TRY(hipSetDevice(h->cfg.device));
    long long jt = ((long long)n * n / (256ll * 2048ll) + 63) / 64 * 64;
    if (jt < 64) jt = 64;
    if (jt > 8192) jt = 8192;
    const int tiles = ceil_div(n, RANK_TILE);
    const int quads = (h->P + 3) / 4, P4 = 4 * quads;
    const int chunks = ceil_div(n, ES_CHUNK);
    const bool sharded = comm != nullptr;
    const bool count_rank = n <= RANK_SORT_MIN;                        // counting rank, keys formed inside the count
    // ses_run_generations with the fitness exchange fused into its producer and consumer: no gathered vector exists, the
    // own values are at h->fit_own and every use of fitness[i] below is for an own row
    const bool fused_fit = comm != nullptr && h->fit_gv != nullptr && !count_rank;
    const bool fused_cnt = h->fit_gv != nullptr && count_rank;          // the counting rank polls the granules (k_rank_count_granules)
    float *const fitness_all = (fused_cnt && comm == nullptr) ? const_cast<float *>(fitness) : nullptr;   // replicated: the kernel writes it
    if (fused_fit || (fused_cnt && comm != nullptr)) fitness = h->fit_own - first_row;
    const int n_own = sharded ? n_rows : n;                            // rows this rank ranks and accumulates
    const int first = sharded ? (int)first_row : 0;
    const int cl = sharded ? per_rank / ES_CHUNK : chunks;             // chunks per rank's payload
    const int stride = sharded ? (cl * P4 + cl + 3) / 4 * 4 : 0;       // floats per payload: cl x P4 partials, cl candidates
    // scratch: sorted tiles (replicated sort path) | ranks of the own rows | local partials / payload | gathered payloads |
    // ticket counters
    const size_t sorted_bytes = (!sharded && !count_rank) ? sizeof(unsigned long long) * (size_t)tiles * RANK_TILE : 0;
    const size_t rank_bytes = (sizeof(int32_t) * (size_t)(sharded ? per_rank : n) + 255) / 256 * 256;
    const size_t local_bytes = (sizeof(float) * (size_t)(sharded ? stride : chunks * P4) + 255) / 256 * 256;
    const size_t gathered_bytes = sharded ? (sizeof(float) * (size_t)stride * world + 255) / 256 * 256 : 0;
    const int rc = ensure_reduce_scratch(h, sorted_bytes + rank_bytes + local_bytes + gathered_bytes + sizeof(unsigned int) * (size_t)quads);
    if (rc != SES_OK) return rc;
    unsigned long long *sorted = (unsigned long long *)h->red_scratch;
    int32_t *rank = (int32_t *)((char *)h->red_scratch + sorted_bytes);
    float *partial = (float *)((char *)rank + rank_bytes);
    float *gathered = (float *)((char *)partial + local_bytes);
    unsigned int *counter = (unsigned int *)((char *)gathered + gathered_bytes);
    // The rank vector is zero on entry: cleared by the perturbation kernel at the end of the previous call, by a memset the
    // first time (or whenever the scratch moved, or the layout / population size changed).
    if (n_own > 0 && (h->rank_zeroed != rank || h->rank_zeroed_n != n_own)) {
        SES_HIP_TRY(hipMemsetAsync(rank, 0, sizeof(int32_t) * (size_t)n_own, h->stream));
    }
    // From here to the launch that clears it again the vector holds counts: every early return below (a failed
    // exchange, a launch error) leaves the cache saying "not zero", so the next call starts with the memset.
    h->rank_zeroed = nullptr;
    double uf = lr / ((double)n * sigma);            // offspring_strategies.py:406-408
    uf *= -1.0;
    const bool final_in_grad = !sharded && chunks <= h->tune_es_final_max_chunks;   // Adam by the gradient kernel's finishing workgroups
    if (n_own > 0) {
        if (fused_cnt) {
            hipLaunchKernelGGL(k_rank_count_granules, dim3(ceil_div(n_own, 256), ceil_div(n, RANK_EP_JT_MAX)), dim3(256), 0, h->stream,
                               *h->fit_gv, h->fit_per_rank, n, RANK_EP_JT_MAX, first, n_own, rank, fitness_all);
        } else if (count_rank && !sharded && h->mean_src && jt <= RANK_EP_JT_MAX) {
            // the episode mean inside the count (ses_run_generations): writes fitness[] for the kernels below and the caller
            hipLaunchKernelGGL(k_rank_count_episodes, dim3(ceil_div(n, 256), ceil_div(n, jt)), dim3(256), 0, h->stream, h->mean_src,
                               h->cfg.eval_ep_num, n, (int)jt, rank, const_cast<float *>(fitness), h->mean_stamp);
        } else if (count_rank) {
            hipLaunchKernelGGL(k_rank_count_fitness, dim3(ceil_div(n_own, 256), ceil_div(n, jt)), dim3(256), 0, h->stream, fitness,
                               n, (int)jt, first, n_own, rank);
        } else if (sharded && fused_fit) {
            hipLaunchKernelGGL((k_rank_sort_search<true>), dim3(ceil_div(n_own, RANK_TILE), tiles), dim3(RANK_TILE / 2), 0, h->stream,
                               h->fit_own, n, first, n_own, rank, *h->fit_gv, per_rank);
        } else if (sharded) {
            hipLaunchKernelGGL((k_rank_sort_search<false>), dim3(ceil_div(n_own, RANK_TILE), tiles), dim3(RANK_TILE / 2), 0, h->stream,
                               fitness, n, first, n_own, rank);
        } else {
            hipLaunchKernelGGL(k_rank_tile_sort, dim3(tiles), dim3(RANK_TILE / 2), 0, h->stream, fitness, n, sorted);
            hipLaunchKernelGGL(k_rank_search, dim3(ceil_div(n, 256), tiles), dim3(256), 0, h->stream, fitness, sorted, n, rank);
        }
    }
    if (final_in_grad) {
        // the gradient kernel's finishing workgroups apply the update; the next launch perturbs the new mu
        if (h->counter_armed != counter) {
            SES_HIP_TRY(hipMemsetAsync(counter, 0, sizeof(unsigned int) * (size_t)quads, h->stream));
            h->counter_armed = counter;
        }
        hipLaunchKernelGGL((k_es_grad_partial_ranked<true>), dim3(quads, chunks), dim3(256), 0, h->stream, rank, fitness, n, 1,
                           seed, gen, P4, partial, best, counter, chunks, h->P, (float)uf, adam_a, mu_in, m_in, v_in, mu_out,
                           m_out, v_out, 0, n, (uint32_t *)nullptr);
    } else {
        // a separate small launch finishes the update; this layout's partial[] may lie over the ticket counter of a
        // smaller population's layout, so a cached "counter is zero" no longer holds
        h->counter_armed = nullptr;
        P2pGranuleView gv;
        // (only while the polling update kernel is a small grid -- P <= 1024: every policy but the GRU ones.  Its workgroups
        //  spin until the peers' granules are there; ranks that SHARE a GPU -- the test rigs -- must not fill it with waiters
        //  and starve the producers they wait for.  Across GPUs there is no such coupling, but one rule serves both.)
        const int grc = (sharded && h->tune_openai_granules && quads <= 256) ? comm_p2p_granules_begin(comm, cl * P4 + cl, &gv)
                                                                              : SES_ERR_UNSUPPORTED;
        if (sharded && grc == SES_ERR_COMM) return grc;
        if (sharded && grc == SES_OK) {
            // peer stores straight from the gradient kernel, the update polls: no launch between them.  Every chunk of the
            // slot is written, also the ones past the end of a ragged last shard (zeros, no candidate)
            hipLaunchKernelGGL((k_es_grad_partial_ranked<false, true>), dim3(quads, cl), dim3(256), 0, h->stream, rank, fitness, n, 1,
                               seed, gen, P4, partial, (float *)nullptr, counter, chunks, h->P, (float)uf, adam_a, mu_in, m_in,
                               v_in, mu_out, m_out, v_out, first, first + n_own, (uint32_t *)nullptr, gv, cl);
            hipLaunchKernelGGL(k_es_apply_granules, dim3(ceil_div(h->P, 4)), dim3(256), 0, h->stream, gv, chunks, h->P, P4,
                               (float)uf, adam_a, mu_in, m_in, v_in, mu_out, m_out, v_out, cl, best);
        } else if (sharded) {
            // (RCCL, or a payload beyond a mailbox section: the partials are all-gathered as floats by a launch of their own)
            hipLaunchKernelGGL((k_es_grad_partial_ranked<false>), dim3(quads, cl), dim3(256), 0, h->stream, rank, fitness, n, 1,
                               seed, gen, P4, partial, (float *)nullptr, counter, chunks, h->P, (float)uf, adam_a, mu_in, m_in,
                               v_in, mu_out, m_out, v_out, first, first + n_own, (uint32_t *)(partial + (size_t)cl * P4));
            SES_HIP_TRY(hipGetLastError());
            const int arc = ses_allgather_fitness(comm, partial, stride, gathered);
            if (arc != SES_OK) return arc;
            hipLaunchKernelGGL(k_es_apply, dim3(ceil_div(h->P, 4)), dim3(256), 0, h->stream, gathered, chunks, h->P, P4,
                               (float)uf, adam_a, mu_in, m_in, v_in, mu_out, m_out, v_out, (float *)nullptr, cl, stride,
                               world * cl, best);
        } else {
            hipLaunchKernelGGL((k_es_grad_partial_ranked<false>), dim3(quads, chunks), dim3(256), 0, h->stream, rank, fitness, n, 1,
                               seed, gen, P4, partial, best, counter, chunks, h->P, (float)uf, adam_a, mu_in, m_in, v_in, mu_out,
                               m_out, v_out, 0, n, (uint32_t *)nullptr);
            if (h->tune_fused_apply_perturb && h->P <= APPLY_PERTURB_MAX_P && chunks <= APPLY_PERTURB_MAX_CHUNKS && n_rows > 0) {
                // the update inside the launch that perturbs the new mean (k_es_apply_perturb)
                const long long threads = (long long)n_rows * quads;
                hipLaunchKernelGGL(k_es_apply_perturb, dim3(ceil_div(threads, 256)), dim3(256), 0, h->stream, partial, chunks, P4,
                                   (float)uf, adam_a, mu_in, m_in, v_in, mu_out, m_out, v_out, next_sigma, seed, next_gen,
                                   (long long)first_row, n_rows, h->P, quads, theta_next, h->stamp, rank, n_own);
                SES_HIP_TRY(hipGetLastError());
                h->rank_zeroed = rank;
                h->rank_zeroed_n = n_own;
                return SES_OK;
            }
            hipLaunchKernelGGL(k_es_apply, dim3(ceil_div(h->P, 4)), dim3(256), 0, h->stream, partial, chunks, h->P, P4,
                               (float)uf, adam_a, mu_in, m_in, v_in, mu_out, m_out, v_out, (float *)nullptr, chunks, 0, 0,
                               (float *)nullptr);
        }
    }
    // the next population from the new mu; the launch also clears the rank vector for the next generation
    {
        const long long threads = (long long)(n_rows > 0 ? n_rows : 1) * quads;
        hipLaunchKernelGGL(k_perturb_openai, dim3(ceil_div(threads, 256)), dim3(256), 0, h->stream, mu_out, next_sigma, seed,
                           next_gen, (long long)first_row, n_rows, h->P, quads, theta_next, h->stamp, rank, n_own);
    }
    SES_HIP_TRY(hipGetLastError());
    h->rank_zeroed = rank;              // cleared by the launch above
    h->rank_zeroed_n = n_own;
    return SES_OK;
}

int ses_openai_generation(ses_handle *h, const float *fitness, int32_t n, uint64_t seed, uint64_t gen, double lr,
                          double sigma, double adam_a, const float *mu_in, const float *m_in, const float *v_in,
                          float *mu_out, float *m_out, float *v_out, float next_sigma, uint64_t next_gen,
                          int64_t first_row, int32_t n_rows, float *theta_next, float *best)
{
    SES_REQUIRE(h && fitness && mu_in && m_in && v_in && mu_out && m_out && v_out, "ses_openai_generation: null argument");
    SES_REQUIRE(mu_in != mu_out && m_in != m_out && v_in != v_out, "ses_openai_generation: in and out vectors must be distinct buffers");
    SES_REQUIRE(n >= 2 && sigma != 0.0, "ses_openai_generation: bad n / sigma");
    SES_REQUIRE(n_rows >= 0 && first_row >= 0 && first_row + n_rows <= (int64_t)n && (n_rows == 0 || theta_next),
                "ses_openai_generation: shard rows [%lld, +%d) outside the population of %d", (long long)first_row, n_rows, n);
    return openai_generation_impl(h, nullptr, fitness, n, seed, gen, lr, sigma, adam_a, mu_in, m_in, v_in, mu_out, m_out, v_out,
                                  next_sigma, next_gen, first_row, n_rows, 0, 1, theta_next, best);
}

// what the all-gather on `comm` can carry: (world of the transport that takes `floats` per rank) or 0
static int comm_world_for(ses_handle *comm, int floats)
{
    int32_t pw = 0, cap = 0, rw = 0;
    if (ses_comm_p2p_info(comm, &pw, &cap, nullptr) != SES_OK || ses_comm_info(comm, nullptr, &rw, nullptr) != SES_OK) return 0;
    if (pw > 0 && floats <= cap && !(comm->tune_comm_force_rccl && rw > 0)) return pw;
    return rw;
}

int ses_openai_sharded_ok(ses_handle *h, ses_handle *comm, int32_t n, int32_t per_rank, int32_t world)
{
    SES_REQUIRE(h && comm, "ses_openai_sharded_ok: null handle");
    if (!h->tune_openai_sharded_tail || n < 2 || world < 2 || per_rank < ES_CHUNK || per_rank % ES_CHUNK != 0) return 0;
    // Round 6: the shard form costs a second exchange, and below 8192 rows in total the replicated tail is four ~5 us launches
    // whatever part of it a rank skips: 4096 rows in total (BASELINE's metric as written) measured 21.5 / 17.5 us replicated against
    // 20.2 / 19.5 us in shard form at 2 / 4 ranks, exchanges between ranks that SHARE a GPU, i.e. without the xGMI flight
    // (profiles/r06_time_tail_strong.txt); from 8192 rows the shard form wins (24.8 against 29.0 us at 2 x 4096).
    if (n < h->tune_openai_sharded_min_rows) return 0;
    if ((long long)per_rank * (world - 1) >= n || (long long)per_rank * world < n) return 0;   // per_rank = ceil(n / world): every rank owns rows
    if (comm->stream != h->stream || comm->cfg.device != h->cfg.device) return 0;
    const int quads = (h->P + 3) / 4, cl = per_rank / ES_CHUNK;
    const int stride = (cl * 4 * quads + cl + 3) / 4 * 4;
    return comm_world_for(comm, stride) == world ? 1 : 0;
}

int ses_openai_generation_sharded(ses_handle *h, ses_handle *comm, const float *fitness, int32_t n, uint64_t seed, uint64_t gen,
                                  double lr, double sigma, double adam_a, const float *mu_in, const float *m_in,
                                  const float *v_in, float *mu_out, float *m_out, float *v_out, float next_sigma,
                                  uint64_t next_gen, int64_t first_row, int32_t n_rows, int32_t per_rank, int32_t world,
                                  float *theta_next, float *best)
{
    SES_REQUIRE(h && comm && fitness && mu_in && m_in && v_in && mu_out && m_out && v_out, "ses_openai_generation_sharded: null argument");
    SES_REQUIRE(mu_in != mu_out && m_in != m_out && v_in != v_out, "ses_openai_generation_sharded: in and out vectors must be distinct buffers");
    SES_REQUIRE(n >= 2 && sigma != 0.0, "ses_openai_generation_sharded: bad n / sigma");
    const int ok = ses_openai_sharded_ok(h, comm, n, per_rank, world);
    if (ok < 0) return ok;
    if (!ok)
        return set_error(SES_ERR_UNSUPPORTED, "ses_openai_generation_sharded: %d rows as %d shards of %d are not chunk-aligned "
                         "(%d rows), or `comm` has no transport of %d ranks for the payload on this stream: use "
                         "ses_openai_generation (ses_openai_sharded_ok tells)", n, world, per_rank, ES_CHUNK, world);
    SES_REQUIRE(first_row >= 0 && first_row % per_rank == 0 && first_row < (int64_t)n &&
                    n_rows == (int32_t)((int64_t)n - first_row < per_rank ? (int64_t)n - first_row : per_rank) && theta_next,
                "ses_openai_generation_sharded: rows [%lld, +%d) are not a rank's shard of %d x %d", (long long)first_row, n_rows,
                world, per_rank);
    return openai_generation_impl(h, comm, fitness, n, seed, gen, lr, sigma, adam_a, mu_in, m_in, v_in, mu_out, m_out, v_out,
                                  next_sigma, next_gen, first_row, n_rows, per_rank, world, theta_next, best);
}

int ses_es_update_stored(ses_handle *h, const double *weights, int32_t n, const float *eps_store, double lr,
                         double sigma, double adam_a, float *mu, float *m, float *v, float *grad_out)
{
    SES_REQUIRE(h && weights && eps_store && mu && m && v, "ses_es_update_stored: null argument");
    SES_REQUIRE(n >= 1 && sigma != 0.0, "ses_es_update_stored: bad n / sigma");
    SES_HIP_TRY(hipSetDevice(h->cfg.device));
    double uf = lr / ((double)n * sigma);
    uf *= -1.0;
    hipLaunchKernelGGL(k_es_update_stored, dim3(ceil_div(h->P, 64)), dim3(64), 0, h->stream, weights, n, eps_store,
                       h->P, (float)uf, adam_a, mu, m, v, grad_out);
    SES_HIP_TRY(hipGetLastError());
    return SES_OK;
}

int ses_elite_ids(ses_handle *h, const int32_t *rank, int32_t n, int32_t k, int32_t *elite_ids)
{
    SES_REQUIRE(h && rank && elite_ids, "ses_elite_ids: null argument");
    SES_REQUIRE(n >= 1 && k >= 1 && k <= n, "ses_elite_ids: need 1 <= k <= n");
    SES_HIP_TRY(hipSetDevice(h->cfg.device));
    hipLaunchKernelGGL(k_elite_ids, dim3(ceil_div(n, 256)), dim3(256), 0, h->stream, rank, n, k, elite_ids);
    SES_HIP_TRY(hipGetLastError());
    return SES_OK;
}

int ses_elite_select(ses_handle *h, const int32_t *rank, int32_t n, int32_t k, const int32_t *parent_map,
                     int32_t *alias_state, int32_t *elite_ids, int32_t *elite_parent_idx, int32_t *alias_first)
{
    SES_REQUIRE(h && rank && parent_map && elite_ids && elite_parent_idx, "ses_elite_select: null argument");
    SES_REQUIRE((alias_state == nullptr) == (alias_first == nullptr),
                "ses_elite_select: alias_state and alias_first go together");
    SES_REQUIRE(n >= 1 && k >= 1 && k <= n && k <= 1024, "ses_elite_select: need 1 <= k <= min(n, 1024)");
    SES_HIP_TRY(hipSetDevice(h->cfg.device));
    hipLaunchKernelGGL(k_elite_select, dim3(1), dim3(1024), 0, h->stream, rank, n, k, parent_map, alias_state, elite_ids,
                       elite_parent_idx, alias_first);
    SES_HIP_TRY(hipGetLastError());
    return SES_OK;
}

int ses_elite_mean(ses_handle *h, const float *rows, const int32_t *alias_first, int32_t k, float *mean)
{
    SES_REQUIRE(h && rows && mean, "ses_elite_mean: null argument");
    SES_REQUIRE(k >= 1, "ses_elite_mean: k must be >= 1");
    SES_HIP_TRY(hipSetDevice(h->cfg.device));
    hipLaunchKernelGGL(k_elite_mean, dim3(ceil_div(h->P, 256)), dim3(256), 0, h->stream, rows, alias_first, k, h->P,
                       mean);
    SES_HIP_TRY(hipGetLastError());
    return SES_OK;
}

int ses_gather_rows(ses_handle *h, const float *src, const int32_t *ids, int32_t n_ids, float *dst)
{
    SES_REQUIRE(h && src && ids && dst, "ses_gather_rows: null argument");
    SES_REQUIRE(n_ids >= 1, "ses_gather_rows: n_ids must be >= 1");
    SES_HIP_TRY(hipSetDevice(h->cfg.device));
    const long long threads = (long long)n_ids * h->P;
    hipLaunchKernelGGL(k_gather_rows, dim3(ceil_div(threads, 256)), dim3(256), 0, h->stream, src, ids, n_ids, h->P,
                       dst);
    SES_HIP_TRY(hipGetLastError());
    return SES_OK;
}

}  // extern "C"
