/*
 * ses_oracle.c -- TEST INFRASTRUCTURE (oracle). Not part of the product path.
 *
 * Plain-C CPU restatement of the simple-es population rollout + fitness hot
 * path, one offspring / one env at a time, in the CANONICAL arithmetic order
 * that the HIP kernels in simple-es_amd/csrc reproduce bit-for-bit.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library.
 *
 * What it restates (file:line into /root/reference):
 *   - GymEnvModel.forward            networks/neural_network.py:20-36
 *   - GymEnvModel.reset (h = 0)      networks/neural_network.py:38-40
 *   - flat parameter order           networks/neural_network.py:46-56 (parameters() order)
 *   - RolloutWorker                  learning_strategies/evolution/loop.py:108-125
 *   - GymWrapper.step truncation     envs/gym_wrapper.py:32-45  (curr_step >= max_step or d)
 *   - CartPolePOMDP / LunarLanderPOMDP obs masks   envs/gym_wrapper.py:57-77
 *   - offspring perturbation theta = mu + sigma*eps   offspring_strategies.py:312-326
 *     (noise source here is Philox4x32-10, the device generator; the reference's
 *      MT19937 stream is reproduced by the numpy oracle in oracle/strategies_np.py)
 *
 * Pinning: tests/test_oracle_golden.py checks this file against fixtures that
 * were produced by importing the reference itself (tests/golden/make_golden.py).
 * CartPole physics is third-party (gym, absent from /root/reference and from
 * this image): "parity unpinned" at that boundary -- the equations below are the
 * published classic-control ones (Barto, Sutton & Anderson 1983 / Florian 2007,
 * as shipped in gym's cartpole.py), evaluated in fp32.
 *
 * Build: see oracle/Makefile (gcc -O2 -mfma -ffp-contract=off).
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include "ses_oracle_math.h"

#define SES_H 32           /* hidden width, hard-coded in neural_network.py:12-17 */
#define SES_MAX_S 32
#define SES_MAX_A 8

/* env ids (shared numbering with include/ses.h) */
#define ENV_CARTPOLE 0

/* rollout modes */
#define MODE_EPISODIC 0     /* env frozen after done (reference semantics)        */
#define MODE_FIXED_LENGTH 1 /* env keeps stepping after done, reward gated by alive */

int o_param_count(int S, int A, int gru)
{
    int p = SES_H * S + SES_H + A * SES_H + A;
    if (gru) p += 2 * (3 * SES_H * SES_H) + 2 * (3 * SES_H);
    return p;
}

/* ------------------------------------------------------------------------- */
/* Philox4x32-10 (Salmon et al., SC'11), same keying as rocRAND's device API:
 * key = seed, counter = {offset/4 lo, offset/4 hi, subsequence lo, subsequence hi}. */
static void philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4])
{
    uint32_t c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3];
    uint32_t k0 = key[0], k1 = key[1];
    for (int r = 0; r < 10; ++r) {
        const uint64_t m0 = (uint64_t)0xD2511F53u * c0;
        const uint64_t m1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t n0 = (uint32_t)(m1 >> 32) ^ c1 ^ k0;
        const uint32_t n1 = (uint32_t)m1;
        const uint32_t n2 = (uint32_t)(m0 >> 32) ^ c3 ^ k1;
        const uint32_t n3 = (uint32_t)m0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

void o_philox_raw(const uint32_t *ctr, const uint32_t *key, uint32_t *out)
{
    philox4x32_10(ctr, key, out);
}

/* stream tags occupy the top byte of the 64-bit subsequence */
#define TAG_PARAM_NOISE 0ull
#define TAG_ENV_INIT 1ull

static void ses_counter(uint64_t tag, uint64_t gen, uint32_t row, uint32_t col, uint32_t ctr[4])
{
    const uint64_t subseq = (tag << 56) | (gen & 0x00FFFFFFFFFFFFFFull);
    ctr[0] = col;
    ctr[1] = row;
    ctr[2] = (uint32_t)subseq;
    ctr[3] = (uint32_t)(subseq >> 32);
}

static inline float u32_to_unit(uint32_t r)
{   /* (r + 0.5) / 2^32 in one rounding; lies in [2^-33, 1] */
    return o_fma((float)r, 0x1.0p-32f, 0x1.0p-33f);
}

/* Box-Muller on two 32-bit words -> two N(0,1) floats */
static inline void box_muller(uint32_t r0, uint32_t r1, float *z0, float *z1)
{
    const float u = u32_to_unit(r0);
    const float ang = o_fma((float)r1, 0x1.921fb6p-30f, 0x1.921fb6p-31f); /* 2pi*(r1+0.5)/2^32 */
    const float rad = sqrtf(-2.0f * o_logf(u));
    float s, c;
    o_sincosf(ang, &s, &c);
    *z0 = rad * c;
    *z1 = rad * s;
}

/* four normals for (seed, gen, offspring row, parameter quad) */
static void normal4(uint64_t seed, uint64_t gen, uint32_t row, uint32_t quad, float z[4])
{
    uint32_t ctr[4], key[2] = {(uint32_t)seed, (uint32_t)(seed >> 32)}, r[4];
    ses_counter(TAG_PARAM_NOISE, gen, row, quad, ctr);
    philox4x32_10(ctr, key, r);
    box_muller(r[0], r[1], &z[0], &z[1]);
    box_muller(r[2], r[3], &z[2], &z[3]);
}

/* eps[n_rows, P] for global offspring rows first_row .. first_row+n_rows-1 */
void o_noise(uint64_t seed, uint64_t gen, int64_t first_row, int n_rows, int P, float *eps)
{
    for (int i = 0; i < n_rows; ++i)
        for (int q = 0; q * 4 < P; ++q) {
            float z[4];
            normal4(seed, gen, (uint32_t)(first_row + i), (uint32_t)q, z);
            for (int l = 0; l < 4 && q * 4 + l < P; ++l) eps[(size_t)i * P + q * 4 + l] = z[l];
        }
}

/*
 * theta[i,:] = parents[k,:] + sigma * eps(seed, gen, first_row+i, :)    if parent_idx[i] = k >= 0
 * theta[i,:] = parents[k,:]  (verbatim copy, no noise)                  if parent_idx[i] = -1-k
 * parent_idx == NULL means "all rows perturb parent 0".
 * Restates offspring_strategies.py:53-60 (genetic), :165-176 (evolution), :310-326 (openai_es);
 * one fp32 fma per element where the reference rounds f64(mu + eps*sigma) to f32 (:322, neural_network.py:55).
 */
void o_perturb(const float *parents, const int32_t *parent_idx, float sigma, uint64_t seed, uint64_t gen,
               int64_t first_row, int n_rows, int P, float *theta)
{
    for (int i = 0; i < n_rows; ++i) {
        const int32_t pi = parent_idx ? parent_idx[i] : 0;
        const float *src = parents + (size_t)(pi >= 0 ? pi : -1 - pi) * P;
        float *dst = theta + (size_t)i * P;
        if (pi < 0) { memcpy(dst, src, sizeof(float) * P); continue; }
        for (int q = 0; q * 4 < P; ++q) {
            float z[4];
            normal4(seed, gen, (uint32_t)(first_row + i), (uint32_t)q, z);
            for (int l = 0; l < 4 && q * 4 + l < P; ++l)
                dst[q * 4 + l] = o_fma(sigma, z[l], src[q * 4 + l]);
        }
    }
}

/* initial env states from the ENV_INIT stream: uniform(-0.05, 0.05), CartPole reset distribution.
 * out[n_rows, E, S]; row = global offspring index (pass shared=1 to key every row as offspring 0). */
void o_init_states_uniform(uint64_t seed, uint64_t gen, int64_t first_row, int n_rows, int E, int S,
                           int shared, float lo, float hi, float *out)
{
    const float span = hi - lo;
    for (int i = 0; i < n_rows; ++i)
        for (int e = 0; e < E; ++e)
            for (int q = 0; q * 4 < S; ++q) {
                uint32_t ctr[4], key[2] = {(uint32_t)seed, (uint32_t)(seed >> 32)}, r[4];
                ses_counter(TAG_ENV_INIT, gen, shared ? 0u : (uint32_t)(first_row + i),
                            (uint32_t)(e * 8 + q), ctr);
                philox4x32_10(ctr, key, r);
                for (int l = 0; l < 4 && q * 4 + l < S; ++l)
                    out[((size_t)i * E + e) * S + q * 4 + l] = o_fma(u32_to_unit(r[l]), span, lo);
            }
}

/* ------------------------------------------------------------------------- */
/* Policy network: canonical evaluation order.                                 */

typedef struct {
    const float *w1, *b1;                 /* fc1 (32,S), (32)            */
    const float *wih, *whh, *bih, *bhh;   /* gru (96,32)x2, (96)x2       */
    const float *w2, *b2;                 /* fc2 (A,32), (A)             */
} net_view;

static net_view view_params(const float *theta, int S, int A, int gru)
{
    net_view v;
    const float *p = theta;
    v.w1 = p; p += SES_H * S;
    v.b1 = p; p += SES_H;
    v.wih = v.whh = v.bih = v.bhh = 0;
    if (gru) {
        v.wih = p; p += 3 * SES_H * SES_H;
        v.whh = p; p += 3 * SES_H * SES_H;
        v.bih = p; p += 3 * SES_H;
        v.bhh = p; p += 3 * SES_H;
    }
    v.w2 = p; p += A * SES_H;
    v.b2 = p;
    return v;
}

/* bias-first, k-ascending fma chain: the numerics of one v_mfma_f32 column / a VALU fma loop */
static inline float dot_chain(const float *w, const float *x, int n, float acc)
{
    for (int k = 0; k < n; ++k) acc = o_fma(w[k], x[k], acc);
    return acc;
}

/* GRU gate row (32 inputs): the device splits the reduction over the two halves of a wavefront, so the
 * canonical order is  (bias + chain over k = 0..15)  +  (0 + chain over k = 16..31). */
static inline float dot_split16(const float *w, const float *x, float bias)
{
    float lo = bias, hi = 0.0f;
    for (int k = 0; k < 16; ++k) lo = o_fma(w[k], x[k], lo);
    for (int k = 16; k < 32; ++k) hi = o_fma(w[k], x[k], hi);
    return lo + hi;
}

/* fc2 row: 8 groups of 4 consecutive hidden units (in-order chain starting from the plain
 * product), then a balanced pairwise tree over the 8 group sums, then + bias. */
static inline float fc2_row(const float *w, const float *h, float bias)
{
    float p[8];
    for (int g = 0; g < 8; ++g) {
        float acc = w[4 * g] * h[4 * g];
        acc = o_fma(w[4 * g + 1], h[4 * g + 1], acc);
        acc = o_fma(w[4 * g + 2], h[4 * g + 2], acc);
        acc = o_fma(w[4 * g + 3], h[4 * g + 3], acc);
        p[g] = acc;
    }
    const float s01 = p[0] + p[1], s23 = p[2] + p[3], s45 = p[4] + p[5], s67 = p[6] + p[7];
    const float s03 = s01 + s23, s47 = s45 + s67;
    return (s03 + s47) + bias;
}

/*
 * One forward pass (neural_network.py:20-36).
 *  obs[S] fp32 (the reference casts obs to fp32 at :22)
 *  h[32]  GRU hidden state, updated in place when gru != 0 (:26)
 *  logits[A] pre-activation fc2 output (:28)
 *  act[A]    tanh(logits) for continuous control (:33)
 * returns argmax(logits) with first-max tie rule for discrete control (:30-31); 0 otherwise.
 */
static int policy_forward(const net_view *v, int S, int A, int discrete, int gru,
                          const float *obs, float *h, float *logits, float *act)
{
    float a[SES_H];
    for (int j = 0; j < SES_H; ++j) a[j] = o_tanhf(dot_chain(v->w1 + j * S, obs, S, v->b1[j]));

    const float *feat = a;
    float hn[SES_H];
    if (gru) {
        /* torch.nn.GRU cell, gate order r, z, n */
        for (int j = 0; j < SES_H; ++j) {
            const float gir = dot_split16(v->wih + (0 * SES_H + j) * SES_H, a, v->bih[0 * SES_H + j]);
            const float giz = dot_split16(v->wih + (1 * SES_H + j) * SES_H, a, v->bih[1 * SES_H + j]);
            const float gin = dot_split16(v->wih + (2 * SES_H + j) * SES_H, a, v->bih[2 * SES_H + j]);
            const float ghr = dot_split16(v->whh + (0 * SES_H + j) * SES_H, h, v->bhh[0 * SES_H + j]);
            const float ghz = dot_split16(v->whh + (1 * SES_H + j) * SES_H, h, v->bhh[1 * SES_H + j]);
            const float ghn = dot_split16(v->whh + (2 * SES_H + j) * SES_H, h, v->bhh[2 * SES_H + j]);
            const float r = o_sigmoidf(gir + ghr);
            const float z = o_sigmoidf(giz + ghz);
            const float n = o_tanhf(o_fma(r, ghn, gin));
            hn[j] = o_fma(z, h[j] - n, n);            /* (1-z)*n + z*h */
        }
        for (int j = 0; j < SES_H; ++j) { h[j] = hn[j]; hn[j] = o_tanhf(hn[j]); }  /* :27 tanh(gru out) */
        feat = hn;
    }

    int best = 0;
    for (int k = 0; k < A; ++k) {
        logits[k] = fc2_row(v->w2 + k * SES_H, feat, v->b2[k]);
        if (act) act[k] = o_tanhf(logits[k]);
        if (logits[k] > logits[best]) best = k;
    }
    return discrete ? best : 0;
}

/* batch entry point: n independent (theta row, obs, h) triples */
void o_policy_forward(int S, int A, int discrete, int gru, int n, const float *theta /*[n,P]*/,
                      const float *obs /*[n,S]*/, float *h /*[n,32] inout or NULL*/,
                      float *logits /*[n,A]*/, float *act /*[n,A] or NULL*/, int32_t *action /*[n]*/)
{
    const int P = o_param_count(S, A, gru);
    float hz[SES_H];
    for (int i = 0; i < n; ++i) {
        net_view v = view_params(theta + (size_t)i * P, S, A, gru);
        float *hp = h ? h + (size_t)i * SES_H : (memset(hz, 0, sizeof hz), hz);
        action[i] = policy_forward(&v, S, A, discrete, gru, obs + (size_t)i * S, hp,
                                   logits + (size_t)i * A, act ? act + (size_t)i * A : 0);
    }
}

/* ------------------------------------------------------------------------- */
/* CartPole-v1 physics, fp32 restatement of the classic-control equations
 * (euler integrator; positions advance with the OLD velocities).              */

#define CP_GRAVITY 9.8f
#define CP_FORCE_OVER_MASS 0x1.22e8bap+3f   /* force_mag / total_mass   = 10 / 1.1             */
#define CP_PML_OVER_MASS 0x1.745d18p-5f     /* polemass_length / total_mass = 0.05 / 1.1        */
#define CP_DEN_C0 0x1.555556p-1f            /* length * 4/3             = 0.5 * 4/3             */
#define CP_DEN_C1 -0x1.745d18p-5f           /* -length * masspole / total_mass = -0.5*0.1/1.1   */
#define CP_TAU 0.02f
#define CP_X_LIMIT 2.4f
#define CP_THETA_LIMIT 0.20943951f /* 12 deg in rad, f32(12*2*pi/360) */
#define CP_CLAMP 1.0e4f            /* state clamp (never active while an episode is alive) */
#define CP_TH_CLAMP 0.75f          /* pole-angle clamp: past-terminal angles saturate at 43 deg (alive: |th| <= 12 deg) */

static inline float clampf(float v, float lim) { return o_minf(o_maxf(v, -lim), lim); }

/* returns 1 when the NEW state is terminal.
 *   temp     = (F + pml*thd^2*sin) / M            with the constant divisions folded into multipliers
 *   thetaacc = (g*sin - cos*temp) / (l*(4/3 - mp*cos^2/M))
 *   xacc     = temp - pml*thetaacc*cos / M                                                        */
static int cartpole_step(float st[4], int action)
{
    const float x = st[0], xd = st[1], th = st[2], thd = st[3];
    const float fom = action == 1 ? CP_FORCE_OVER_MASS : -CP_FORCE_OVER_MASS;
    float s, c;
    o_sincosf(th, &s, &c);
    const float temp = o_fma(CP_PML_OVER_MASS * (thd * thd), s, fom);
    const float num = o_fma(-c, temp, CP_GRAVITY * s);
    const float den = o_fma(CP_DEN_C1, c * c, CP_DEN_C0);
    const float thacc = num / den;
    const float xacc = o_fma(-CP_PML_OVER_MASS * thacc, c, temp);
    st[0] = clampf(o_fma(CP_TAU, xd, x), CP_CLAMP);
    st[1] = clampf(o_fma(CP_TAU, xacc, xd), CP_CLAMP);
    st[2] = clampf(o_fma(CP_TAU, thd, th), CP_TH_CLAMP);
    st[3] = clampf(o_fma(CP_TAU, thacc, thd), CP_CLAMP);
    return (st[0] < -CP_X_LIMIT) | (st[0] > CP_X_LIMIT) | (st[2] < -CP_THETA_LIMIT) | (st[2] > CP_THETA_LIMIT);
}

/* Gym-order float64 CartPole ("physics64"): the statements of gym's cartpole.py step() one by one, every
 * operation a separately rounded IEEE double operation (no fma), with the deterministic o_sincos64 in place of
 * libm.  Observations handed to the policy are float32(state) as in the reference (neural_network.py:22). */
static int cartpole_step64(double st[4], int action)
{
    const double gravity = 9.8, masscart = 1.0, masspole = 0.1, length = 0.5, force_mag = 10.0, tau = 0.02;
    const double total_mass = masspole + masscart, polemass_length = masspole * length;
    const double x = st[0], x_dot = st[1], theta = st[2], theta_dot = st[3];
    const double force = action == 1 ? force_mag : -force_mag;
    double sintheta, costheta;
    o_sincos64(theta, &sintheta, &costheta);
    const double temp = (force + ((polemass_length * (theta_dot * theta_dot)) * sintheta)) / total_mass;
    const double thetaacc = ((gravity * sintheta) - (costheta * temp)) /
                            (length * ((4.0 / 3.0) - ((masspole * (costheta * costheta)) / total_mass)));
    const double xacc = temp - (((polemass_length * thetaacc) * costheta) / total_mass);
    double nx = x + tau * x_dot, nxd = x_dot + tau * xacc, nth = theta + tau * theta_dot, nthd = theta_dot + tau * thetaacc;
    const double lim = 1.0e4;                       /* only reachable after termination (fixed-length mode) */
    nx = nx < -lim ? -lim : (nx > lim ? lim : nx);
    nxd = nxd < -lim ? -lim : (nxd > lim ? lim : nxd);
    nth = nth < -lim ? -lim : (nth > lim ? lim : nth);
    nthd = nthd < -lim ? -lim : (nthd > lim ? lim : nthd);
    st[0] = nx; st[1] = nxd; st[2] = nth; st[3] = nthd;
    const double thr = 12 * 2 * 3.141592653589793 / 360;
    return (nx < -2.4) | (nx > 2.4) | (nth < -thr) | (nth > thr);
}

/*
 * Standalone SoA env step over n envs (the unit the device K3 kernel is checked against).
 * status word: bits 0..30 = steps taken so far, bit 31 = done.
 * A done env is left untouched in MODE_EPISODIC; in MODE_FIXED_LENGTH its physics keeps
 * advancing but neither ret nor the step count change.
 */
void o_cartpole_step_soa(int n, int mode, int max_step, float *x, float *xd, float *th, float *thd,
                         const int32_t *action, float *ret, uint32_t *status)
{
    for (int i = 0; i < n; ++i) {
        const uint32_t stw = status[i];
        const int done = (int)(stw >> 31);
        uint32_t steps = stw & 0x7fffffffu;
        if (done && mode == MODE_EPISODIC) continue;
        float st[4] = {x[i], xd[i], th[i], thd[i]};
        const int term = cartpole_step(st, action[i]);
        x[i] = st[0]; xd[i] = st[1]; th[i] = st[2]; thd[i] = st[3];
        if (!done) {
            ret[i] += 1.0f;                    /* reward 1 on every step incl. the terminal one */
            steps += 1;
            const int now_done = term || (max_step > 0 && (int)steps >= max_step);
            status[i] = steps | ((uint32_t)now_done << 31);
        }
    }
}

/*
 * Whole-population rollout (loop.py:108-125): for each offspring i and episode e, run the
 * policy against its own env copy until done or max_step; fitness[i] = sum_e return / E.
 *   init      [E,S] (init_per_offspring=0, shared by all offspring) or [N,E,S]
 *   obs_mask  bit k set -> observation component k is zeroed before the policy sees it
 *             (CartPolePOMDP = 0b1010, envs/gym_wrapper.py:73-77)
 *   ep_return [N,E] f64 per-episode undiscounted return;  ep_steps [N,E] steps taken
 *   fitness   [N]   f32
 */
static void rollout_cartpole_impl(int physics64, int S, int A, int discrete, int gru, int N, int E, int max_step, int mode,
                                  uint32_t obs_mask, const float *theta, const float *init, int init_per_offspring,
                                  double *ep_return, int32_t *ep_steps, float *fitness);

void o_rollout_cartpole(int S, int A, int discrete, int gru, int N, int E, int max_step, int mode,
                        uint32_t obs_mask, const float *theta, const float *init, int init_per_offspring,
                        double *ep_return, int32_t *ep_steps, float *fitness)
{
    rollout_cartpole_impl(0, S, A, discrete, gru, N, E, max_step, mode, obs_mask, theta, init, init_per_offspring,
                          ep_return, ep_steps, fitness);
}

void o_rollout_cartpole64(int S, int A, int discrete, int gru, int N, int E, int max_step, int mode,
                          uint32_t obs_mask, const float *theta, const float *init, int init_per_offspring,
                          double *ep_return, int32_t *ep_steps, float *fitness)
{
    rollout_cartpole_impl(1, S, A, discrete, gru, N, E, max_step, mode, obs_mask, theta, init, init_per_offspring,
                          ep_return, ep_steps, fitness);
}

static void rollout_cartpole_impl(int physics64, int S, int A, int discrete, int gru, int N, int E, int max_step, int mode,
                                  uint32_t obs_mask, const float *theta, const float *init, int init_per_offspring,
                                  double *ep_return, int32_t *ep_steps, float *fitness)
{
    const int P = o_param_count(S, A, gru);
    for (int i = 0; i < N; ++i) {
        net_view v = view_params(theta + (size_t)i * P, S, A, gru);
        double total = 0.0;
        for (int e = 0; e < E; ++e) {
            const float *s0 = init + ((size_t)(init_per_offspring ? i : 0) * E + e) * 4;
            float st[4] = {s0[0], s0[1], s0[2], s0[3]};
            double st64[4] = {s0[0], s0[1], s0[2], s0[3]};
            float h[SES_H] = {0};
            float logits[SES_MAX_A], act[SES_MAX_A];
            double ret = 0.0;
            int steps = 0, alive = 1;
            for (int t = 0; t < max_step; ++t) {
                if (!alive && mode == MODE_EPISODIC) break;
                float obs[4];
                for (int k = 0; k < 4; ++k) {
                    const float sv = physics64 ? (float)st64[k] : st[k];
                    obs[k] = ((obs_mask >> k) & 1u) ? 0.0f : sv;
                }
                const int a = policy_forward(&v, S, A, discrete, gru, obs, h, logits, act);
                const int term = physics64 ? cartpole_step64(st64, a) : cartpole_step(st, a);
                if (alive) {
                    ret += 1.0;
                    steps += 1;
                    if (term || steps >= max_step) alive = 0;
                }
            }
            ep_return[(size_t)i * E + e] = ret;
            if (ep_steps) ep_steps[(size_t)i * E + e] = steps;
            total += ret;
        }
        fitness[i] = (float)(total / (double)E);
    }
}

/* ========================================================================================== */
/* MPE simple_spread (pettingzoo mpe, reached by the reference through
 * envs/pettingzoo_wrapper.py:9,22-58).  pettingzoo is third-party and absent from the reference tree and
 * from this image: PARITY UNPINNED at this boundary; the rules below follow SURVEY Appendix A.3 and are
 * the build's own definition (fp32).
 *   n agents (size 0.15, colliding, mass 1), n landmarks; dt 0.1, damping 0.25, contact force 1e2,
 *   contact margin 1e-3, action sensitivity 5; 5 discrete actions {noop, -x, +x, -y, +y};
 *   team reward per cycle = sum_i [0.5*global + 0.5*local_i], global = -sum_landmarks min_agents dist,
 *   local_i = -(number of OTHER agents closer than 0.3).
 * State layout of one env: apos[2n] avel[2n] lpos[2n].  Observation of agent i (pettingzoo order):
 *   vel_i(2) pos_i(2) landmark_k - pos_i (2n) agent_j - pos_i for j != i (2(n-1)) comm zeros (2(n-1)). */
#define SP_MAXN 4
#define SP_DT 0.1f
#define SP_DAMP_KEEP 0.75f
#define SP_CONTACT_FORCE 100.0f
#define SP_CONTACT_MARGIN 1.0e-3f
#define SP_INV_MARGIN 1000.0f
#define SP_DIST_MIN 0.3f
#define SP_SENS 5.0f

/* softplus-style penetration: logaddexp(0, y) * k with y = -(dist - dist_min)/k */
static inline float sp_penetration(float dist)
{
    const float y = -(dist - SP_DIST_MIN) * SP_INV_MARGIN;
    const float ay = fabsf(y);
    const float l1p = o_logf(1.0f + o_expf(-ay));       /* log(1 + e^-|y|) */
    return (o_maxf(y, 0.0f) + l1p) * SP_CONTACT_MARGIN;
}

static void spread_obs(int n, const float *st, int i, float *obs)
{
    const float *ap = st, *av = st + 2 * n, *lp = st + 4 * n;
    int o = 0;
    obs[o++] = av[2 * i]; obs[o++] = av[2 * i + 1];
    obs[o++] = ap[2 * i]; obs[o++] = ap[2 * i + 1];
    for (int k = 0; k < n; ++k) { obs[o++] = lp[2 * k] - ap[2 * i]; obs[o++] = lp[2 * k + 1] - ap[2 * i + 1]; }
    for (int j = 0; j < n; ++j) if (j != i) { obs[o++] = ap[2 * j] - ap[2 * i]; obs[o++] = ap[2 * j + 1] - ap[2 * i + 1]; }
    for (int j = 0; j < n - 1; ++j) { obs[o++] = 0.0f; obs[o++] = 0.0f; }
}

/* one world step with the agents' discrete actions; returns the team reward of this cycle */
static float spread_step(int n, float *st, const int *action)
{
    float *ap = st, *av = st + 2 * n;
    const float *lp = st + 4 * n;
    float fx[SP_MAXN], fy[SP_MAXN];
    for (int i = 0; i < n; ++i) {
        const int a = action[i];
        fx[i] = a == 1 ? -SP_SENS : (a == 2 ? SP_SENS : 0.0f);
        fy[i] = a == 3 ? -SP_SENS : (a == 4 ? SP_SENS : 0.0f);
    }
    for (int a = 0; a < n; ++a)
        for (int b = a + 1; b < n; ++b) {
            const float dx = ap[2 * a] - ap[2 * b], dy = ap[2 * a + 1] - ap[2 * b + 1];
            const float dist = sqrtf(o_fma(dx, dx, dy * dy));
            const float pen = sp_penetration(dist);
            const float scale = dist > 0.0f ? (SP_CONTACT_FORCE * pen) / dist : 0.0f;
            const float gx = dx * scale, gy = dy * scale;
            fx[a] = gx + fx[a]; fy[a] = gy + fy[a];
            fx[b] = fx[b] - gx; fy[b] = fy[b] - gy;
        }
    for (int i = 0; i < n; ++i) {
        const float vx = o_fma(fx[i], SP_DT, av[2 * i] * SP_DAMP_KEEP);
        const float vy = o_fma(fy[i], SP_DT, av[2 * i + 1] * SP_DAMP_KEEP);
        av[2 * i] = vx; av[2 * i + 1] = vy;
        ap[2 * i] = o_fma(vx, SP_DT, ap[2 * i]);
        ap[2 * i + 1] = o_fma(vy, SP_DT, ap[2 * i + 1]);
    }
    float global = 0.0f;
    for (int k = 0; k < n; ++k) {
        float best = 0.0f;
        for (int i = 0; i < n; ++i) {
            const float dx = ap[2 * i] - lp[2 * k], dy = ap[2 * i + 1] - lp[2 * k + 1];
            const float d = sqrtf(o_fma(dx, dx, dy * dy));
            best = i == 0 ? d : o_minf(best, d);
        }
        global = global - best;
    }
    float team = 0.0f;
    for (int i = 0; i < n; ++i) {
        float local = 0.0f;
        for (int j = 0; j < n; ++j) if (j != i) {
            const float dx = ap[2 * i] - ap[2 * j], dy = ap[2 * i + 1] - ap[2 * j + 1];
            const float d = sqrtf(o_fma(dx, dx, dy * dy));
            if (d < SP_DIST_MIN) local = local - 1.0f;
        }
        team = team + o_fma(0.5f, global, 0.5f * local);
    }
    return team;
}

/* single-env helpers for the Python env object */
void o_spread_obs(int n, const float *st, int i, float *obs) { spread_obs(n, st, i, obs); }
float o_spread_step(int n, float *st, const int32_t *action) { return spread_step(n, st, (const int *)action); }

/*
 * Population rollout for simple_spread: every agent of a team runs the SAME offspring network
 * (learning_strategies/evolution/utils.py:4-8) on its own observation; team return summed over cycles
 * (pettingzoo_wrapper.py:45-52, loop.py:123).  init: [E, 4n] (agent positions then landmark positions;
 * velocities start at 0) shared, or [N, E, 4n].  S = 6n, A = 5, MLP policy.
 */
void o_rollout_spread(int n_agents, int N, int E, int max_cycles, const float *theta, const float *init,
                      int init_per_offspring, double *ep_return, float *fitness)
{
    const int n = n_agents, S = 6 * n, A = 5;
    const int P = o_param_count(S, A, 0);
    for (int i = 0; i < N; ++i) {
        net_view v = view_params(theta + (size_t)i * P, S, A, 0);
        double total = 0.0;
        for (int e = 0; e < E; ++e) {
            const float *s0 = init + ((size_t)(init_per_offspring ? i : 0) * E + e) * 4 * n;
            float st[6 * SP_MAXN];
            for (int k = 0; k < 2 * n; ++k) { st[k] = s0[k]; st[2 * n + k] = 0.0f; st[4 * n + k] = s0[2 * n + k]; }
            double ret = 0.0;
            for (int t = 0; t < max_cycles; ++t) {
                int action[SP_MAXN];
                for (int a = 0; a < n; ++a) {
                    float obs[6 * SP_MAXN], logits[SES_MAX_A];
                    spread_obs(n, st, a, obs);
                    action[a] = policy_forward(&v, S, A, 1, 0, obs, 0, logits, 0);
                }
                ret += (double)spread_step(n, st, action);
            }
            ep_return[(size_t)i * E + e] = ret;
            total += ret;
        }
        fitness[i] = (float)(total / (double)E);
    }
}

/* ========================================================================================== */
/* LunarLanderContinuous-v2 (conf/lunarlander_openai.yaml of the reference: GRU policy, 4 tanh outputs of
 * which the env uses [0] main and [1] side engine, POMDP mask on obs 2,3,5 -- envs/gym_wrapper.py:57-66).
 *
 * The env itself -- gym's lunar_lander.py on a Box2D world of three bodies, two revolute joints and polygon /
 * terrain-edge contacts, world.Step(1/50, 180, 60) per env step -- is restated in ses_b2.h (the world) and
 * ses_lander_env.h (the env), built as C++ in ses_b2_oracle.cpp; this file drives it through the C entry points
 * below.  gym / Box2D are third-party, absent from the reference tree and from this image: PARITY UNPINNED at that
 * boundary (the headers list what is restated and what deviates).  The GPU kernels reproduce THIS definition bit
 * for bit (tests/test_gpu_lander.py).
 * One initial-state row = 16 uniforms in [0,1): [0,1] initial force, [2..13] terrain heights, [14,15] the
 * episode's dispersion-noise key (bit patterns). */
int o_lander_state_size(void);
void o_lander_reset(void *state, const float *u16, float *obs);
float o_lander_step(void *state, float a0, float a1, float *obs, int32_t *done);
void o_lander_obs(const void *state, float *obs);

/* population rollout: GRU or MLP policy with 4 continuous (tanh) outputs, POMDP mask bits over the 8 obs */
void o_rollout_lander(int gru, int N, int E, int max_step, uint32_t obs_mask, const float *theta, const float *init,
                      int init_per_offspring, double *ep_return, int32_t *ep_steps, float *fitness)
{
    const int S = 8, A = 4;
    const int P = o_param_count(S, A, gru);
    void *st = malloc((size_t)o_lander_state_size());
    for (int i = 0; i < N; ++i) {
        net_view v = view_params(theta + (size_t)i * P, S, A, gru);
        double total = 0.0;
        for (int e = 0; e < E; ++e) {
            float obs[8];
            o_lander_reset(st, init + ((size_t)(init_per_offspring ? i : 0) * E + e) * 16, obs);
            float h[SES_H] = {0};
            double ret = 0.0;
            int steps = 0;
            int32_t done = 0;
            while (!done && steps < max_step) {
                float logits[SES_MAX_A], act[SES_MAX_A];
                for (int k = 0; k < 8; ++k) if ((obs_mask >> k) & 1u) obs[k] = 0.0f;
                (void)policy_forward(&v, S, A, 0, gru, obs, h, logits, act);
                ret += (double)o_lander_step(st, act[0], act[1], obs, &done);
                steps += 1;
            }
            ep_return[(size_t)i * E + e] = ret;
            if (ep_steps) ep_steps[(size_t)i * E + e] = steps;
            total += ret;
        }
        fitness[i] = (float)(total / (double)E);
    }
    free(st);
}

/* ========================================================================================== */
/* BipedalWalker-v3 (conf/bipedalwalker.yaml of the reference: MLP policy, 24 observations, 4 tanh outputs = the four
 * joint motors).  The env -- gym's bipedal_walker.py on a Box2D world of five bodies and four revolute joints -- is
 * restated in ses_b2.h / ses_walker_env.h (C++, ses_b2_oracle.cpp); parity with gym / Box2D is UNPINNED (see there).
 * One initial-state row = 4 floats: [0] uniform of the initial hull force, [1], [2] bit patterns of the key of the
 * episode's terrain stream, [3] unused. */
int o_walker_state_size(void);
void o_walker_reset(void *state, const float *u4, float *obs);
float o_walker_step(void *state, const float *action4, float *obs, int32_t *done);

void o_rollout_walker(int gru, int N, int E, int max_step, const float *theta, const float *init,
                      int init_per_offspring, double *ep_return, int32_t *ep_steps, float *fitness)
{
    const int S = 24, A = 4;
    const int P = o_param_count(S, A, gru);
    void *st = malloc((size_t)o_walker_state_size());
    for (int i = 0; i < N; ++i) {
        net_view v = view_params(theta + (size_t)i * P, S, A, gru);
        double total = 0.0;
        for (int e = 0; e < E; ++e) {
            float obs[24];
            o_walker_reset(st, init + ((size_t)(init_per_offspring ? i : 0) * E + e) * 4, obs);
            float h[SES_H] = {0};
            double ret = 0.0;
            int steps = 0;
            int32_t done = 0;
            while (!done && steps < max_step) {
                float logits[SES_MAX_A], act[SES_MAX_A];
                (void)policy_forward(&v, S, A, 0, gru, obs, h, logits, act);
                ret += (double)o_walker_step(st, act, obs, &done);
                steps += 1;
            }
            ep_return[(size_t)i * E + e] = ret;
            if (ep_steps) ep_steps[(size_t)i * E + e] = steps;
            total += ret;
        }
        fitness[i] = (float)(total / (double)E);
    }
    free(st);
}
