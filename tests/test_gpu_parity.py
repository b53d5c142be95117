"""HIP path vs oracle on a real MI355X -- every call goes through the C ABI (ses._lib / ses.device).

Bars (written per test):
  * perturbation noise, policy forward, env step, rollout returns, ranks, elite mean, the stored-noise
    ES update: BIT-EXACT against the C / numpy oracle on the same seeded inputs.
  * Philox-mode ES gradient: fp32 reduction in a different (tree) order -> rtol 2e-5 of max|grad|.
  * golden fixtures produced by the imported reference: returns within 1e-4 (north_star tolerance).
"""
import json
import os

import numpy as np
import pytest
import torch

from oracle import c_oracle as co
from oracle import strategies_np as snp

pytestmark = pytest.mark.gpu

RETURN_TOL = 1e-4


@pytest.fixture(scope="module")
def es():
    from ses import HipES
    h = HipES("CartPole-v1", 4, 2, True, False, pomdp=False, max_step=500, eval_ep_num=5)
    yield h
    h.close()


def dev(a, dtype=None):
    t = torch.from_numpy(np.ascontiguousarray(a))
    if dtype is not None:
        t = t.to(dtype)
    return t.cuda()


def host(t):
    return t.detach().cpu().numpy()


def bits(a):
    return np.ascontiguousarray(a).view(np.uint32 if a.dtype.itemsize == 4 else np.uint64)


def assert_bit_equal(got, want, what):
    got, want = np.asarray(got), np.asarray(want)
    assert got.shape == want.shape, (what, got.shape, want.shape)
    bad = bits(got) != bits(want)
    assert not bad.any(), f"{what}: {bad.sum()} of {bad.size} elements differ, first at {np.argwhere(bad)[0]}: " \
                          f"{got[tuple(np.argwhere(bad)[0])]!r} vs {want[tuple(np.argwhere(bad)[0])]!r}"


# ----------------------------------------------------------------------------------------- K1
def test_noise_bit_exact_and_rocrand_is_philox(es):
    """rocRAND's device Philox engine + our Box-Muller == the oracle's Random123 restatement."""
    for seed, gen, first, n in ((0, 0, 0, 64), (12345678901234, 7, 1000, 33), (2 ** 63 + 5, 2 ** 40 + 3, 2 ** 29, 8)):
        got = host(es.noise(seed, gen, first, n))
        assert_bit_equal(got, co.noise(seed, gen, first, n, es.P), f"noise seed={seed}")
    big = host(es.noise(1, 1, 0, 4096))
    assert abs(big.mean()) < 3e-3 and abs(big.std() - 1) < 3e-3


def test_perturb_bit_exact(es):
    rng = np.random.RandomState(0)
    parents = rng.randn(3, es.P).astype(np.float32)
    idx = np.array([-1, -3, 0, 1, 2, 2, 0, -2] * 4, np.int32)
    n = idx.size
    got = host(es.perturb(dev(parents), 0.37, 9, 4, 100, n, parent_idx=dev(idx)))
    assert_bit_equal(got, co.perturb(parents, idx, 0.37, 9, 4, 100, n), "perturb(parent map)")
    # default map: every row perturbs parent 0; shards of the same generation tile the full matrix
    full = host(es.perturb(dev(parents[:1]), 0.1, 5, 2, 0, 96))
    part = host(es.perturb(dev(parents[:1]), 0.1, 5, 2, 64, 32))
    assert_bit_equal(full, co.perturb(parents[:1], None, 0.1, 5, 2, 0, 96), "perturb")
    assert_bit_equal(part, full[64:], "perturb shard")
    # explicit row ids (used to regenerate elites on any rank)
    rows = np.array([5, 70, 3, 95], np.int32)
    sel = host(es.perturb(dev(parents[:1]), 0.1, 5, 2, 0, 4, row_ids=dev(rows)))
    assert_bit_equal(sel, full[rows], "perturb row_ids")


def test_perturb_host_noise_matches_reference_arithmetic(es):
    """float32(float64(mu) + eps64*sigma) -- offspring_strategies.py:321-322, bit-exact."""
    rng = np.random.RandomState(1)
    mu = rng.randn(1, es.P).astype(np.float32)
    eps = rng.standard_normal((17, es.P))
    sigma = 0.168
    theta, store = es.perturb_host_noise(dev(mu), dev(eps), sigma, want_eps_store=True)
    want = mu.copy().repeat(17, 0)
    want += eps * sigma
    want_store = mu.copy().repeat(17, 0)
    want_store += eps
    assert_bit_equal(host(theta), want, "theta")
    assert_bit_equal(host(store), want_store, "eps_store")


def test_init_states_bit_exact(es):
    got = host(es.init_states_uniform(3, 9, 10, 7))
    assert_bit_equal(got, co.init_states_uniform(3, 9, 10, 7, 5, 4, False), "init states")
    sh = host(es.init_states_uniform(3, 9, 10, 7, shared=True))
    assert_bit_equal(sh, co.init_states_uniform(3, 9, 10, 7, 5, 4, True), "shared init states")


# ----------------------------------------------------------------------------------------- K2
@pytest.mark.parametrize("S,A,disc", [(4, 2, True), (8, 4, False), (12, 5, True), (18, 5, True)])
def test_policy_forward_bit_exact(S, A, disc):
    from ses import HipES
    h = HipES(None, S, A, disc, False)
    rng = np.random.RandomState(S * 10 + A)
    n = 777
    theta = (rng.randn(n, h.P) * rng.choice([0.1, 0.5, 2.0], size=(n, 1))).astype(np.float32)
    obs = (rng.randn(n, S) * rng.choice([0.05, 1.0, 4.0], size=(n, 1))).astype(np.float32)
    obs[:5] = 0.0
    action, logits, act = h.policy_forward(dev(theta), dev(obs))
    o_action, o_logits, o_act, _ = co.policy_forward(S, A, disc, False, theta, obs)
    assert_bit_equal(host(logits), o_logits, "logits")
    assert_bit_equal(host(act), o_act, "tanh(logits)")
    if disc:
        assert np.array_equal(host(action), o_action)
    h.close()


def test_tanh_special_arguments_on_the_hardware():
    """The table tanh at the arguments where an instruction's corner case could separate the device from the CPU build:
    interval boundaries and their neighbours (v_fract_f32 / v_cvt_i32_f32 of the same value must agree), zeros,
    denormals, the clamp at 10, infinities and NaN (v_min_f32 absorbs it).  Probed through a network that routes one
    input straight into one hidden unit and that unit straight into logit 0: logits[0] == tanh(x), act[0] ==
    tanh(tanh(x)); also exactly odd, which the reference's exact argmax ties rely on."""
    from ses import HipES
    S, A = 4, 2
    h = HipES(None, S, A, False, False)
    grid = np.arange(-10.5, 10.5, 1 / 32, dtype=np.float32)
    xs = np.concatenate([grid, np.nextafter(grid, np.float32(-20)), np.nextafter(grid, np.float32(20)),
                         np.float32([0.0, -0.0, -1e-45, 1e-45, -1e-38, -1e-30, -1e-10, -3e-9, -2e-8, 1e-10, 9.9999990463, -9.9999990463,
                                     10.0, -10.0, 50.0, -50.0, 1e30, -1e30, np.inf, -np.inf, np.nan]),
                         np.random.RandomState(0).randn(4000).astype(np.float32) * 3]).astype(np.float32)
    n = xs.shape[0]
    theta = np.zeros((n, h.P), np.float32)
    theta[:, 5 * S + 0] = 1.0                           # W1[5][0] = 1: hidden unit 5 sees obs[0]
    theta[:, 32 * S + 32 + 5] = 1.0                     # W2[0][5] = 1: logit 0 sees hidden unit 5
    obs = np.zeros((n, S), np.float32)
    obs[:, 0] = xs
    _, logits, act = h.policy_forward(dev(theta), dev(obs))
    _, o_logits, o_act, _ = co.policy_forward(S, A, False, False, theta, obs)
    assert_bit_equal(host(logits), o_logits, "tanh(x)")
    assert_bit_equal(host(act), o_act, "tanh(tanh(x))")
    finite = np.isfinite(xs)
    assert np.abs(o_logits[finite, 0] - np.tanh(xs[finite].astype(np.float64))).max() <= 1.2e-7
    obs[:, 0] = -xs
    _, neg_logits, _ = h.policy_forward(dev(theta), dev(obs))
    assert np.array_equal(host(neg_logits)[finite, 0], -o_logits[finite, 0])   # as values: the logit's sum of zeros turns -0 into +0
    h.close()


def test_policy_forward_golden_g1(golden_dir):
    """Device forward against the REFERENCE's own outputs (fixture G1), fp32 tolerance."""
    from ses import HipES
    g1 = np.load(os.path.join(golden_dir, "g1_forward.npz"))
    for ci in (0, 3, 4, 5):
        S, A, disc, gru = (int(v) for v in g1[f"c{ci}_cfg"])
        h = HipES(None, S, A, bool(disc), False)
        theta, obs = g1[f"c{ci}_theta"], g1[f"c{ci}_obs"]
        nets, T, _ = obs.shape
        th_rep = np.repeat(theta, T, axis=0)
        action, logits, act = h.policy_forward(dev(th_rep), dev(obs.reshape(nets * T, S)))
        np.testing.assert_allclose(host(logits).reshape(nets, T, A), g1[f"c{ci}_logits"], rtol=2e-6, atol=2e-5)
        if disc:
            assert np.array_equal(host(action).reshape(nets, T), g1[f"c{ci}_act"][:, :, 0].astype(np.int32))
        else:
            np.testing.assert_allclose(host(act).reshape(nets, T, A), g1[f"c{ci}_act"], rtol=0, atol=3e-6)
        h.close()


# ----------------------------------------------------------------------------------------- K3
@pytest.mark.parametrize("mode", [0, 1])
@pytest.mark.parametrize("n", [1, 3, 4, 1000, 4099, 65536 + 5])
def test_env_step_bit_exact(es, n, mode):
    rng = np.random.RandomState(n + mode)
    st = [rng.uniform(-0.3, 0.3, n).astype(np.float32) for _ in range(4)]
    st[0] = rng.uniform(-2.6, 2.6, n).astype(np.float32)       # some beyond the x limit
    ret = np.zeros(n, np.float32)
    status = np.zeros(n, np.uint32)
    d = [dev(a) for a in st]
    d_ret, d_status = dev(ret), dev(status.view(np.int32))
    for t in range(12):
        action = rng.randint(0, 2, n).astype(np.int32)
        co.cartpole_step_soa(mode, 8, st[0], st[1], st[2], st[3], action, ret, status)
        # handle was created with max_step=500; use a dedicated one for truncation at 8
        es_t = _trunc_handle()
        es_t.env_step(d[0], d[1], d[2], d[3], dev(action), d_ret, d_status, mode=mode)
    for k in range(4):
        assert_bit_equal(host(d[k]), st[k], f"state[{k}]")
    assert_bit_equal(host(d_ret), ret, "ret")
    assert np.array_equal(host(d_status).view(np.uint32), status)
    assert (status >> 31).all()                                 # everything truncated at 8 steps at the latest


_TRUNC = {}


def _trunc_handle():
    from ses import HipES
    if "h" not in _TRUNC:
        _TRUNC["h"] = HipES("CartPole-v1", 4, 2, True, False, max_step=8, eval_ep_num=1)
    return _TRUNC["h"]


@pytest.mark.parametrize("mode", [0, 1])
def test_env_step_wild_states_general_sincos_and_angle_clamp(es, mode):
    """Caller-supplied states far outside the live region: pole angles beyond pi/4 take the general sin/cos
    (argument reduction) instead of the small-angle form, and past-terminal angles saturate at 0.75 rad."""
    n = 20000
    rng = np.random.RandomState(77 + mode)
    st = [rng.uniform(-3, 3, n).astype(np.float32), rng.uniform(-20, 20, n).astype(np.float32),
          rng.uniform(-40, 40, n).astype(np.float32), rng.uniform(-60, 60, n).astype(np.float32)]
    st[2][::7] = rng.uniform(-0.2, 0.2, st[2][::7].size).astype(np.float32)     # waves with mixed small / large angles
    ret, status = np.zeros(n, np.float32), np.zeros(n, np.uint32)
    d = [dev(a) for a in st]
    d_ret, d_status = dev(ret), dev(status.view(np.int32))
    for t in range(6):
        action = rng.randint(0, 2, n).astype(np.int32)
        co.cartpole_step_soa(mode, 500, st[0], st[1], st[2], st[3], action, ret, status)
        es.env_step(d[0], d[1], d[2], d[3], dev(action), d_ret, d_status, mode=mode)
    for k in range(4):
        assert_bit_equal(host(d[k]), st[k], f"state[{k}]")
    assert_bit_equal(host(d_ret), ret, "ret")
    if mode == 1:
        assert np.abs(st[2]).max() == np.float32(0.75)           # the clamp was active


def test_env_step_default_shape_is_derived_from_the_device():
    """The default reservation is a seventh of the device's LDS per CU (no constant in the source): the occupancy calculator
    must see 7 single-wave workgroups per CU for it, and the waves knob must move it."""
    from ses import HipES
    h = HipES("CartPole-v1", 4, 2, True, False, max_step=500, eval_ep_num=1)
    block, lds, wpc, per_cu = h.env_step_shape()
    assert block == 64 and wpc == 7 and per_cu >= 65536, (block, lds, wpc, per_cu)
    assert per_cu // 8 < lds <= per_cu // 7, (lds, per_cu)
    h.set_tuning("env_step_waves_per_cu", 5)
    assert h.env_step_shape()[2] == 5 and per_cu // 6 < h.env_step_shape()[1] <= per_cu // 5
    h.set_tuning("env_step_lds_bytes", 0)
    h.set_tuning("env_step_block", 256)
    assert h.env_step_shape()[1] == 0 and h.env_step_shape()[2] >= 8
    h.close()


@pytest.mark.parametrize("block,lds", [(64, -1), (256, 0), (128, 40960), (64, 10240), (256, 65536)])
def test_env_step_launch_shapes_change_no_bit(block, lds):
    """ses_env_step limits its waves in flight with an LDS reservation (knobs env_step_block / env_step_lds_bytes): every
    shape -- the derived default (-1), the old unlimited one -- returns the bits of the oracle."""
    from ses import HipES
    n = 70000 + 4 * 37                                            # several workgroups of every shape, a ragged tail
    rng = np.random.RandomState(block + lds)
    st = [rng.uniform(-0.3, 0.3, n).astype(np.float32) for _ in range(4)]
    ret, status = np.zeros(n, np.float32), np.zeros(n, np.uint32)
    h = HipES("CartPole-v1", 4, 2, True, False, max_step=500, eval_ep_num=1)
    h.set_tuning("env_step_block", block)
    h.set_tuning("env_step_lds_bytes", lds)
    d = [dev(a) for a in st]
    d_ret, d_status = dev(ret), dev(status.view(np.int32))
    for t in range(5):
        action = rng.randint(0, 2, n).astype(np.int32)
        co.cartpole_step_soa(1, 500, st[0], st[1], st[2], st[3], action, ret, status)
        h.env_step(d[0], d[1], d[2], d[3], dev(action), d_ret, d_status, mode=1)
    for k in range(4):
        assert_bit_equal(host(d[k]), st[k], f"state[{k}]")
    assert_bit_equal(host(d_ret), ret, "ret")
    assert np.array_equal(host(d_status).view(np.uint32), status)
    h.close()


def test_env_step_unaligned_views_take_scalar_path(es):
    n = 1001
    rng = np.random.RandomState(3)
    base = [rng.uniform(-0.05, 0.05, n + 1).astype(np.float32) for _ in range(4)]
    st = [b[1:].copy() for b in base]
    d = [dev(b)[1:] for b in base]                               # 4-byte offset: not 16-B aligned
    ret, status = np.zeros(n, np.float32), np.zeros(n, np.uint32)
    d_ret, d_status = dev(ret), dev(status.view(np.int32))
    action = rng.randint(0, 2, n).astype(np.int32)
    co.cartpole_step_soa(0, 500, st[0], st[1], st[2], st[3], action, ret, status)
    es.env_step(d[0], d[1], d[2], d[3], dev(action), d_ret, d_status)
    for k in range(4):
        assert_bit_equal(host(d[k]), st[k], f"state[{k}]")


# ----------------------------------------------------------------------------------------- rollout
@pytest.mark.parametrize("lpe", [0, 1, 2, 4, 8, 16, 32])
def test_rollout_golden_g5_and_oracle(golden_dir, lpe):
    """Fixture G5: returns of the REFERENCE RolloutWorker + GymEnvModel (torch) over the build's fp32
    CartPole.  Device returns must be within 1e-4 of them and bit-equal to the C oracle, for every
    lanes-per-env variant and in both rollout modes."""
    from ses import HipES
    g = np.load(os.path.join(golden_dir, "g56_rollouts.npz"))
    theta, init = g["g5_theta"], g["init_states"]
    h = HipES("CartPole-v1", 4, 2, True, False, max_step=500, eval_ep_num=5, lanes_per_env=lpe)
    o_fit, o_ret, o_steps = co.rollout_cartpole(theta, init, 5, 500)
    # (8 / 16 lanes per env and the library's own choice: the scalar step and the packed step of lone waves, ses_policy_pk.h)
    for mode, packed in [(m, k) for m in (0, 1) for k in ((0, 1) if lpe in (0, 8, 16) else (0,))]:
        h.set_tuning("rollout_packed", packed)
        fit, ep_ret, ep_steps = h.rollout(dev(theta), dev(init), mode=mode, want_episodes=True)
        assert np.array_equal(host(ep_steps), o_steps), f"mode {mode}: episode lengths differ from the oracle"
        assert_bit_equal(host(fit), o_fit, "fitness")
        assert np.array_equal(host(ep_ret), o_ret)
        assert np.abs(host(fit).astype(np.float64) - g["g5_returns"]).max() <= RETURN_TOL
    h.close()


def test_rollout_per_offspring_init_and_pomdp(es):
    from ses import HipES
    rng = np.random.RandomState(11)
    n = 300
    theta = (rng.randn(n, 226) * 1.0).astype(np.float32)
    init = rng.uniform(-0.05, 0.05, (n, 5, 4)).astype(np.float32)
    fit = host(es.rollout(dev(theta), dev(init)))
    o_fit, _, _ = co.rollout_cartpole(theta, init, 5, 500)
    assert_bit_equal(fit, o_fit, "per-offspring init")
    hp = HipES("CartPole-v1", 4, 2, True, False, pomdp=True, max_step=200, eval_ep_num=5)
    fitp = host(hp.rollout(dev(theta), dev(init)))
    o_fitp, _, _ = co.rollout_cartpole(theta, init, 5, 200, obs_mask=0b1010)
    assert_bit_equal(fitp, o_fitp, "POMDP mask + max_step 200")
    assert not np.array_equal(fitp, np.minimum(fit, 200))
    hp.close()


@pytest.mark.parametrize("n", [2048, 1700, 2457, 3000, 4096])
def test_rollout_mixed_splits_equal_the_oracle(n):
    """Populations between one and a few waves per SIMD run MIXED splits of the lanes: 8 lanes per env on every SIMD + the rest at
    16 (2048 offspring x 5 episodes, round 6), 16 lanes + the rest at 4 (the headline's 4096 x 5).  Every split -- the library's
    choice, the (8, 16) mix switched off, the mixes switched off altogether -- is the same canonical arithmetic: bit-equal to the
    oracle in both modes, ragged last waves included (1700 x 5 = 8500 envs: 77 waves of the second kind, the last one a quarter full)."""
    from ses import HipES
    rng = np.random.RandomState(n)
    theta = (rng.randn(n, 226) * 0.6).astype(np.float32)
    init = rng.uniform(-0.05, 0.05, (n, 5, 4)).astype(np.float32)
    o_fit, _, o_steps = co.rollout_cartpole(theta, init, 5, 40)
    for knobs in ({}, {"rollout_mix_8_16": 0}, {"rollout_mix": 0}):
        h = HipES("CartPole-v1", 4, 2, True, False, max_step=40, eval_ep_num=5)
        for name, value in knobs.items():
            h.set_tuning(name, value)
        for mode in (0, 1):
            fit, _, ep_steps = h.rollout(dev(theta), dev(init), mode=mode, want_episodes=True)
            assert np.array_equal(host(ep_steps), o_steps), (knobs, mode)
            assert_bit_equal(host(fit), o_fit, f"{knobs} mode={mode}")
        h.close()


@pytest.mark.parametrize("lpe", [0, 4, 8, 16, 32])
def test_rollout_wild_initial_states_take_the_general_loop(lpe):
    """Initial pole angles outside |th| <= 0.78 (not a reset the env produces, but the ABI accepts any state):
    such waves run the loop with the full sin/cos argument reduction; waves of ordinary resets run the
    small-angle loop.  Both must agree with the oracle bit for bit, in both modes, with and without a mask."""
    from ses import HipES
    rng = np.random.RandomState(5)
    n = 640
    theta = (rng.randn(n, 226) * 0.8).astype(np.float32)
    init = rng.uniform(-0.05, 0.05, (n, 5, 4)).astype(np.float32)
    wild = rng.rand(n) < 0.3                                          # whole offspring, so some waves stay ordinary
    init[wild, :, 2] = rng.uniform(-3.0, 3.0, (wild.sum(), 5)).astype(np.float32)
    init[wild, :, 3] = (-init[wild, :, 2] / 0.02 + rng.uniform(-5, 5, (wild.sum(), 5))).astype(np.float32)  # swings back
    for pomdp, mask in ((False, 0), (True, 0b1010)):
        h = HipES("CartPole-v1", 4, 2, True, False, pomdp=pomdp, max_step=300, eval_ep_num=5, lanes_per_env=lpe)
        o_fit, _, o_steps = co.rollout_cartpole(theta, init, 5, 300, obs_mask=mask)
        assert (o_steps.reshape(n, 5)[wild] > 1).any()                # some wild starts survive the first step
        for mode, packed in [(m, k) for m in (0, 1) for k in ((0, 1) if lpe in (0, 8, 16) else (0,))]:
            h.set_tuning("rollout_packed", packed)                    # (a packed wave with a wild start falls back to the general loop)
            fit, _, ep_steps = h.rollout(dev(theta), dev(init), mode=mode, want_episodes=True)
            assert np.array_equal(host(ep_steps), o_steps), (pomdp, mode)
            assert_bit_equal(host(fit), o_fit, f"pomdp={pomdp} mode={mode}")
        h.close()


@pytest.mark.parametrize("n", [1638, 1639, 1640, 2049, 9830, 9831])
def test_rollout_split_boundaries(n):
    """Population sizes on both sides of the boundaries of the split chooser (csrc/ses_rollout.hip, choose_cartpole_mlp_split:
    up to 8192 episodes one of the pure splits -- 16, 8 or 4 lanes per env, by the issue-cost model --, from there to 49 152 also
    "one light wave per SIMD at 16 or 8 lanes per env + the rest at 4", pure 4 lanes per env above): the ragged last waves of
    each part must still land on the oracle's bits."""
    from ses import HipES
    rng = np.random.RandomState(n)
    theta = (rng.randn(n, 226) * 0.6).astype(np.float32)
    init = rng.uniform(-0.05, 0.05, (n, 5, 4)).astype(np.float32)
    h = HipES("CartPole-v1", 4, 2, True, False, max_step=40, eval_ep_num=5)
    o_fit, _, o_steps = co.rollout_cartpole(theta, init, 5, 40)
    for mode in (0, 1):
        fit, _, ep_steps = h.rollout(dev(theta), dev(init), mode=mode, want_episodes=True)
        assert np.array_equal(host(ep_steps), o_steps), (n, mode)
        assert_bit_equal(host(fit), o_fit, f"n={n} mode={mode}")
    h.close()


def test_rollout_edge_sizes(es):
    """1 offspring, ragged tail of a wave, zero network (action 0 forever)."""
    init = np.random.RandomState(0).uniform(-0.05, 0.05, (5, 4)).astype(np.float32)
    for n in (1, 2, 13, 65):
        theta = (np.random.RandomState(n).randn(n, 226) * 0.7).astype(np.float32)
        assert_bit_equal(host(es.rollout(dev(theta), dev(init))), co.rollout_cartpole(theta, init, 5, 500)[0], f"n={n}")
    zero = np.zeros((4, 226), np.float32)
    fit = host(es.rollout(dev(zero), dev(init)))
    assert_bit_equal(fit, co.rollout_cartpole(zero, init, 5, 500)[0], "zero network")
    assert fit.max() < 20                                        # pushing left forever falls over quickly


def test_rollout_full_size_properties(es):
    """BASELINE.json configs[1] size (4096 x 5 episodes x <=500 steps): size-independent properties --
    sharding invariance, mode invariance, permutation equivariance -- plus an oracle spot check."""
    rng = np.random.RandomState(5)
    n = 4096
    theta = (rng.randn(n, 226) * 0.5).astype(np.float32)
    init = rng.uniform(-0.05, 0.05, (5, 4)).astype(np.float32)
    d_theta, d_init = dev(theta), dev(init)
    fit = host(es.rollout(d_theta, d_init))
    assert (fit >= 1).all() and (fit <= 500).all() and np.allclose(fit * 5, np.round(fit * 5))
    assert_bit_equal(host(es.rollout(d_theta, d_init, mode=1)), fit, "fixed-length mode")
    halves = np.concatenate([host(es.rollout(d_theta[:1000].contiguous(), d_init)),
                             host(es.rollout(d_theta[1000:].contiguous(), d_init))])
    assert_bit_equal(halves, fit, "sharded rollout")
    perm = rng.permutation(n)
    assert_bit_equal(host(es.rollout(dev(theta[perm]), d_init)), fit[perm], "permuted population")
    sample = rng.choice(n, 256, replace=False)
    assert_bit_equal(fit[sample], co.rollout_cartpole(theta[sample], init, 5, 500)[0], "oracle spot check")


# ----------------------------------------------------------------------------------------- K4-K6
@pytest.mark.parametrize("n", [2, 16, 97, 1025, 4096, 8192, 8193, 20001, 65536])
def test_rank_center(es, n):
    rng = np.random.RandomState(n)
    fit = rng.permutation(n).astype(np.float32) * 0.5 + 3            # tie-free
    rank, w = es.rank_center(dev(fit))
    order = snp.rank_desc(list(fit))
    want_rank = np.empty(n, np.int32)
    want_rank[order] = np.arange(n)
    assert np.array_equal(host(rank), want_rank)
    np.testing.assert_allclose(host(w), snp.centered_ranks(list(fit)), rtol=0, atol=1e-12)
    # massive ties (CartPole returns saturate at 500): stable rule = reward desc, index desc
    tied = rng.choice([500.0, 10.0, 9.8, 137.2], n).astype(np.float32)
    rank_t, w_t = es.rank_center(dev(tied))
    order_t = snp.rank_desc(list(tied), stable=True)
    want_t = np.empty(n, np.int32)
    want_t[order_t] = np.arange(n)
    assert np.array_equal(host(rank_t), want_t)
    assert sorted(host(rank_t)) == list(range(n))
    np.testing.assert_allclose(host(w_t), snp.centered_ranks(list(tied), stable=True), rtol=0, atol=1e-12)


def test_es_update_stored_is_bit_exact_with_reference_arithmetic(es, golden_dir):
    """Fixture G3 (reference openai_es.evaluate incl. Adam): device update from the reference's own
    epsilons matrix must reproduce mu, m, v bit for bit."""
    g = np.load(os.path.join(golden_dir, "g234_strategies.npz"))
    meta = json.load(open(os.path.join(golden_dir, "g234_strategies.json")))["es_mlp"]
    n, lr, sigma0, decay = 16, 0.05, 0.1, 0.999
    mu, m, v = es.zeros(es.P), es.zeros(es.P), es.zeros(es.P)
    adam = snp.AdamNP(np.zeros(es.P, np.float32), lr)
    eps_store = np.zeros((n, es.P), np.float32)
    # generation 0 epsilons: reconstruct from the fixture's theta0 is lossy; use eps{g} for g >= 1 and
    # replay generation 0 with the numpy oracle (np.random.seed(7)) to get its epsilons matrix
    np.random.seed(meta["seed"])
    strat = snp.OpenAIESNP(es.P, sigma0, decay, lr, n)
    sigma = sigma0
    for gen in range(meta["gens"]):
        eps_store = np.stack(strat.epsilons)
        rewards = g[f"es_mlp_rewards{gen}"]
        rank, w = es.rank_center(dev(rewards.astype(np.float32)))
        adam.t += 1
        a = adam.step_scale()
        es.es_update_stored(w, dev(eps_store), lr, sigma, a, mu, m, v)
        strat.evaluate(list(rewards))
        sigma *= decay
        assert_bit_equal(host(mu), g[f"es_mlp_mu{gen + 1}"], f"mu after generation {gen + 1}")
        assert_bit_equal(host(m), g[f"es_mlp_m{gen + 1}"], "adam m")
        assert_bit_equal(host(v), g[f"es_mlp_v{gen + 1}"], "adam v")


def test_es_update_philox_matches_numpy_oracle(es):
    n, lr, sigma, seed, gen = 512, 0.05, 0.1, 21, 6
    rng = np.random.RandomState(2)
    fit = rng.permutation(n).astype(np.float32)
    rank, w = es.rank_center(dev(fit))
    mu0 = rng.randn(es.P).astype(np.float32) * 0.3
    m0 = rng.randn(es.P).astype(np.float32) * 1e-3
    v0 = (rng.rand(es.P).astype(np.float32) * 1e-5)
    mu, m, v = dev(mu0), dev(m0), dev(v0)
    adam = snp.AdamNP(mu0.copy(), lr)
    adam.m, adam.v, adam.t = m0.copy(), v0.copy(), 3
    adam.t += 1
    a = adam.step_scale()
    adam.t -= 1
    grad = host(es.es_update_philox(w, seed, gen, lr, sigma, a, mu, m, v, skip_row0=True, want_grad=True))
    eps = co.noise(seed, gen, 0, n, es.P).astype(np.float64)
    eps[0] = 0.0
    wn = host(w)
    want = (wn[:, None] * eps).sum(0) * (-lr / (n * sigma))
    np.testing.assert_allclose(grad, want, rtol=0, atol=2e-5 * np.abs(want).max())
    adam.update(grad)                                             # Adam itself is exact given the same grad
    assert_bit_equal(host(mu), adam.theta, "mu")
    assert_bit_equal(host(m), adam.m, "m")
    assert_bit_equal(host(v), adam.v, "v")
    # determinism: a second call from the same state gives the same bits
    mu2, m2, v2 = dev(mu0), dev(m0), dev(v0)
    grad2 = host(es.es_update_philox(w, seed, gen, lr, sigma, a, mu2, m2, v2, skip_row0=True, want_grad=True))
    assert_bit_equal(grad2, grad, "grad determinism")


def test_elite_selection_and_mean(es):
    rng = np.random.RandomState(8)
    n, k = 97, 10
    theta = rng.randn(n, es.P).astype(np.float32)
    fit = rng.permutation(n).astype(np.float32)
    rank, _ = es.rank_center(dev(fit))
    ids = host(es.elite_ids(rank, k))
    assert np.array_equal(ids, snp.rank_desc(list(fit))[:k])
    rows = es.gather_rows(dev(theta), dev(ids.astype(np.int32)))
    assert_bit_equal(host(rows), theta[ids], "gather")
    want = theta[ids[0]].copy()
    for j in ids[1:]:
        want += theta[j]
    want /= k
    assert_bit_equal(host(es.elite_mean(rows)), want, "elite mean")
    # aliasing quirk: elite j is the same object as elite 0 -> running sum doubles
    alias = np.zeros(k, np.int32)
    alias[3] = 1
    want = theta[ids[0]].copy()
    for pos, j in enumerate(ids[1:], start=1):
        want += want if alias[pos] else theta[j]
    want /= k
    assert_bit_equal(host(es.elite_mean(rows, dev(alias))), want, "elite mean with alias")


@pytest.mark.parametrize("n,k", [(2, 1), (17, 5), (97, 10), (4097, 64), (65536, 1024)])
def test_elite_select_matches_host_bookkeeping(es, n, k):
    """ses_elite_select = elite ids + parent-map gather + the aliasing flags / state of simple_evolution
    (offspring_strategies.py:234-248), checked against the host logic it replaces over a sequence of generations."""
    rng = np.random.RandomState(n + k)
    parent_map = rng.randint(-2, 3, n).astype(np.int32)
    state_dev = dev(np.ones(1, np.int32))
    state = True
    for gen in range(6):
        rank = rng.permutation(n).astype(np.int32)
        if gen % 2 == 0 and n > 2:                       # make slots 0 and 1 elites often enough to exercise the flags
            a, b = np.where(rank == 0)[0][0], np.where(rank == min(1, k - 1))[0][0]
            rank[[0, a]] = rank[[a, 0]]
            if k > 1:
                b = np.where(rank == 1)[0][0]
                rank[[1, b]] = rank[[b, 1]]
        ids, sel, alias = es.elite_select(dev(rank), k, dev(parent_map), state_dev)
        want_ids = np.empty(k, np.int32)
        for i in range(n):
            if rank[i] < k:
                want_ids[rank[i]] = i
        want_alias = np.zeros(k, np.int32)
        if state and want_ids[0] in (0, 1):
            for j in range(1, k):
                if want_ids[j] in (0, 1) and want_ids[j] != want_ids[0]:
                    want_alias[j] = 1
        state = bool(want_ids[0] == 0 or (want_ids[0] == 1 and state))
        assert np.array_equal(host(ids), want_ids)
        assert np.array_equal(host(sel), parent_map[want_ids])
        assert np.array_equal(host(alias), want_alias)
        assert int(host(state_dev)[0]) == int(state)
    ids2, sel2, alias2 = es.elite_select(dev(rank), k, dev(parent_map))       # simple_genetic form: no alias outputs
    assert alias2 is None and np.array_equal(host(ids2), want_ids) and np.array_equal(host(sel2), parent_map[want_ids])


def test_bad_arguments_raise_before_launch(es):
    from ses import SesError
    with pytest.raises(SesError):
        es.rollout(torch.zeros(4, 225, device="cuda"), torch.zeros(5, 4, device="cuda"))
    with pytest.raises(SesError):
        es.rollout(torch.zeros(4, 226, device="cuda"), torch.zeros(4, 4, device="cuda"))
    with pytest.raises(SesError):
        es.rollout(torch.zeros(4, 226), torch.zeros(5, 4))                  # CPU tensors
    with pytest.raises(SesError):
        es.rank_center(torch.zeros(1, device="cuda"))
    with pytest.raises(SesError):
        es.gather_rows(torch.zeros(4, 226, device="cuda"), torch.tensor([0, 4], dtype=torch.int32, device="cuda"))


def test_raw_c_abi_error_paths(es):
    """Straight through ctypes, no Python-side validation: bad arguments must come back as negative status codes
    with a message -- never a launch."""
    import ctypes
    from ses import _lib
    lib = _lib.load()
    h = es._h
    assert lib.ses_rollout(h, None, None, 0, 4, 0, None, None, None) == -1
    assert b"null argument" in lib.ses_last_error()
    t = torch.zeros(8, 226, device="cuda")
    i = torch.zeros(5, 4, device="cuda")
    f = torch.zeros(8, device="cuda")
    p = lambda x: ctypes.c_void_p(x.data_ptr())
    assert lib.ses_rollout(h, p(t), p(i), 0, 0, 0, p(f), None, None) == -1            # n_rows = 0
    assert lib.ses_rollout(h, p(t), p(i), 0, 8, 7, p(f), None, None) == -1            # unknown mode
    assert b"bad mode" in lib.ses_last_error()
    r = torch.zeros(1, dtype=torch.int32, device="cuda")
    assert lib.ses_rank_center(h, p(f), 1, p(r), None, None) == -1                          # n - 1 = 0 divides in the reference
    assert lib.ses_elite_ids(h, p(r), 1, 2, p(r)) == -1                               # k > n
    bad = _lib.SesConfig(0, 5, 2, 1, 0, 0, 500, 5, 0, 0, 1, 0)                           # CartPole with num_state 5
    out = ctypes.c_void_p()
    assert lib.ses_create(ctypes.byref(bad), None, ctypes.byref(out)) == -1 and not out.value
    bad = _lib.SesConfig(0, 4, 2, 1, 0, 0, 500, 5, 99, 0, 1, 0)                          # device out of range
    assert lib.ses_create(ctypes.byref(bad), None, ctypes.byref(out)) == -1
    bad = _lib.SesConfig(7, 4, 2, 1, 0, 0, 500, 5, 0, 0, 1, 0)                           # unknown env
    assert lib.ses_create(ctypes.byref(bad), None, ctypes.byref(out)) == -1
    assert lib.ses_destroy(None) == 0
    # the handle is still healthy afterwards
    fit = es.rollout(torch.zeros(8, 226, device="cuda"), torch.zeros(5, 4, device="cuda"))
    assert fit.shape == (8,)


@pytest.mark.parametrize("gru", [False, True])
def test_physics64_rollout_bit_exact(golden_dir, gru):
    """env.physics = float64: gym-order double-precision CartPole on the device == the C oracle, bit for bit."""
    from ses import HipES
    g = np.load(os.path.join(golden_dir, "g56_rollouts.npz"))
    theta = g["g5gru_theta"] if gru else g["g5_theta"]
    init = g["init_states"]
    h = HipES("CartPole-v1", 4, 2, True, gru, pomdp=gru, max_step=500, eval_ep_num=5, physics64=True)
    o_fit, _, o_steps = co.rollout_cartpole(theta, init, 5, 500, gru=gru, obs_mask=0b1010 if gru else 0, physics64=True)
    for mode in (0, 1):
        fit, _, steps = h.rollout(dev(theta), dev(init), mode=mode, want_episodes=True)
        assert np.array_equal(host(steps), o_steps)
        assert_bit_equal(host(fit), o_fit, "fitness")
    if not gru:
        agree = np.mean(np.abs(o_fit.astype(np.float64) - g["g5_returns_gym64"]) <= RETURN_TOL)
        assert agree >= 0.99
        # a larger population takes the 4-lanes-per-env variant
        rng = np.random.RandomState(9)
        big = (rng.randn(3000, 226) * 0.6).astype(np.float32)
        assert_bit_equal(host(h.rollout(dev(big), dev(init))), co.rollout_cartpole(big, init, 5, 500, physics64=True)[0],
                         "population of 3000")
    h.close()
