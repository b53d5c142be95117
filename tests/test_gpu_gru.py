"""GRU policy (nn.GRU(32,32) cell, networks/neural_network.py:13-27) on the HIP path: the wave-per-offspring
kernels of ses_gru.h against the C oracle (bit-exact) and the reference's own outputs (fixtures G1, G5-gru)."""
import os

import numpy as np
import pytest
import torch

from oracle import c_oracle as co

pytestmark = pytest.mark.gpu
RETURN_TOL = 1e-4


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def bits(a):
    return np.ascontiguousarray(a).view(np.uint32)


@pytest.mark.parametrize("S,A,disc", [(4, 2, True), (8, 4, False)])
def test_gru_forward_bit_exact_over_a_sequence(S, A, disc):
    from ses import HipES
    h = HipES(None, S, A, disc, True)
    rng = np.random.RandomState(S + A)
    n, T = 203, 12
    theta = (rng.randn(n, h.P) * rng.choice([0.1, 0.5, 1.5], size=(n, 1))).astype(np.float32)
    hid_o = np.zeros((n, 32), np.float32)
    hid_d = dev(hid_o.copy())
    d_theta = dev(theta)
    for t in range(T):
        obs = (rng.randn(n, S) * rng.choice([0.05, 1.0, 3.0], size=(n, 1))).astype(np.float32)
        if t == 0:
            obs[:7] = 0.0
        action, logits, act = h.policy_forward(d_theta, dev(obs), hid_d)
        o_action, o_logits, o_act, hid_o = co.policy_forward(S, A, disc, True, theta, obs, hid_o)
        assert np.array_equal(bits(hid_d.cpu().numpy()), bits(hid_o)), f"hidden state differs at step {t}"
        assert np.array_equal(bits(logits.cpu().numpy()), bits(o_logits)), f"logits differ at step {t}"
        assert np.array_equal(bits(act.cpu().numpy()), bits(o_act))
        if disc:
            assert np.array_equal(action.cpu().numpy(), o_action)
    h.close()


@pytest.mark.parametrize("ci", [1, 2, 6])
def test_gru_forward_golden_g1(golden_dir, ci):
    """Teacher-forced single steps against the reference module's outputs (fixture G1)."""
    from ses import HipES
    g1 = np.load(os.path.join(golden_dir, "g1_forward.npz"))
    S, A, disc, gru = (int(v) for v in g1[f"c{ci}_cfg"])
    assert gru
    h = HipES(None, S, A, bool(disc), True)
    theta, obs = g1[f"c{ci}_theta"], g1[f"c{ci}_obs"]
    nets, T, _ = obs.shape
    d_theta = dev(theta)
    h_prev = np.zeros((nets, 32), np.float32)
    for t in range(T):
        hid = dev(h_prev)
        action, logits, act = h.policy_forward(d_theta, dev(obs[:, t]), hid)
        np.testing.assert_allclose(logits.cpu().numpy(), g1[f"c{ci}_logits"][:, t], rtol=2e-6, atol=2e-5)
        np.testing.assert_allclose(hid.cpu().numpy(), g1[f"c{ci}_h"][:, t], rtol=0, atol=1e-5)
        if disc:
            assert np.array_equal(action.cpu().numpy(), g1[f"c{ci}_act"][:, t, 0].astype(np.int32))
        else:
            np.testing.assert_allclose(act.cpu().numpy(), g1[f"c{ci}_act"][:, t], rtol=0, atol=1e-5)
        h_prev = g1[f"c{ci}_h"][:, t].copy()
    h.close()


@pytest.mark.parametrize("pomdp", [True, False])
def test_gru_rollout_bit_exact_and_golden(golden_dir, pomdp):
    from ses import HipES
    g = np.load(os.path.join(golden_dir, "g56_rollouts.npz"))
    theta, init = g["g5gru_theta"], g["init_states"]
    es = HipES("CartPole-v1", 4, 2, True, True, pomdp=pomdp, max_step=500, eval_ep_num=5)
    o_fit, o_ret, o_steps = co.rollout_cartpole(theta, init, 5, 500, gru=True, obs_mask=0b1010 if pomdp else 0)
    for mode in (0, 1):
        fit, ep_ret, ep_steps = es.rollout(dev(theta), dev(init), mode=mode, want_episodes=True)
        assert np.array_equal(ep_steps.cpu().numpy(), o_steps)
        assert np.array_equal(bits(fit.cpu().numpy()), bits(o_fit))
    if pomdp:   # fixture G5-gru: reference RolloutWorker + reference GRU module over the POMDP CartPole
        assert np.abs(o_fit.astype(np.float64) - g["g5gru_returns"]).max() <= RETURN_TOL
    es.close()


def test_gru_rollout_random_population_and_shards():
    from ses import HipES
    rng = np.random.RandomState(3)
    n = 150
    theta = (rng.randn(n, 6562) * 0.4).astype(np.float32)
    init = rng.uniform(-0.05, 0.05, (n, 5, 4)).astype(np.float32)
    es = HipES("CartPole-v1", 4, 2, True, True, pomdp=True, max_step=300, eval_ep_num=5)
    fit = es.rollout(dev(theta), dev(init)).cpu().numpy()
    o_fit, _, _ = co.rollout_cartpole(theta, init, 5, 300, gru=True, obs_mask=0b1010)
    assert np.array_equal(bits(fit), bits(o_fit))
    parts = np.concatenate([es.rollout(dev(theta[:37]), dev(init[:37])).cpu().numpy(),
                            es.rollout(dev(theta[37:]), dev(init[37:])).cpu().numpy()])
    assert np.array_equal(bits(parts), bits(fit))
    es.close()


@pytest.mark.parametrize("E", [12, 16, 20])
def test_gru_rollout_on_the_matrix_cores_bit_exact(E):
    """eval_ep_num >= 12 runs the GRU rollout with v_mfma_f32_16x16x4_f32 (ses_gru_mfma.h): episodes are the tile
    columns (20 = one full batch of 16 + a batch of 4).  A run of MFMAs over ascending k-blocks is the fmaf chain
    of the canonical arithmetic, so returns must still equal the C oracle's bit for bit, in both modes."""
    from ses import HipES
    rng = np.random.RandomState(E)
    n = 4200 // E + 1                                            # > 4096 episodes: past the episode-parallel small-population path
    theta = (rng.randn(n, 6562) * 0.4).astype(np.float32)
    init = rng.uniform(-0.05, 0.05, (n, E, 4)).astype(np.float32)
    for pomdp, mask in ((True, 0b1010), (False, 0)):
        es = HipES("CartPole-v1", 4, 2, True, True, pomdp=pomdp, max_step=200, eval_ep_num=E)
        o_fit, o_ret, o_steps = co.rollout_cartpole(theta, init, E, 200, gru=True, obs_mask=mask)
        for mode in (0, 1):
            fit, ep_ret, ep_steps = es.rollout(dev(theta), dev(init), mode=mode, want_episodes=True)
            assert np.array_equal(ep_steps.cpu().numpy(), o_steps), (E, pomdp, mode)
            assert np.array_equal(bits(fit.cpu().numpy()), bits(o_fit))
        es.close()
    assert o_steps.max() > 50                                   # some policies balance for a while


def test_lander_gru_rollout_on_the_matrix_cores_bit_exact():
    from ses import HipES
    rng = np.random.RandomState(31)
    n, E = 320, 13                                               # 4160 episodes: past the small-population path
    theta = (rng.randn(n, 6756) * 0.3).astype(np.float32)
    init = rng.uniform(0, 1, (n, E, 16)).astype(np.float32)
    es = HipES("LunarLanderContinuous-v2", 8, 4, False, True, pomdp=True, max_step=120, eval_ep_num=E)
    fit, ep_ret, ep_steps = es.rollout(dev(theta), dev(init), want_episodes=True)
    o_fit, o_ret, o_steps = co.rollout_lander(theta, init, E, 120)
    assert np.array_equal(ep_steps.cpu().numpy(), o_steps)
    assert np.array_equal(ep_ret.cpu().numpy(), o_ret)
    assert np.array_equal(bits(fit.cpu().numpy()), bits(o_fit))
    es.close()


@pytest.mark.parametrize("knobs", [{"SES_TUNING": "gru_ep_parallel_max=0,gru_mfma4_min_e=0"},
                                   {"SES_TUNING": "gru_ep_parallel_max=0,gru_mfma_min_e=1"},
                                   {"SES_TUNING": "gru_ep_parallel_max=0,gru_mfma4_min_e=1"},
                                   {"SES_TUNING": "gru_ep_parallel_max=1000000"},
                                   {"SES_TUNING": "gru_ep_parallel_max=0,lander_offspring_per_wave=2"},
                                   {"SES_TUNING": "gru_ep_parallel_max=0,lander_offspring_per_wave=4"}],
                         ids=["lockstep", "mfma", "mfma_4x4x1", "episode_parallel", "lander_2_per_wave", "lander_4_per_wave"])
def test_gru_parity_suites_on_every_kernel_path(knobs):
    """ses_rollout picks the GRU kernel from the population size and episode count: one wave per (offspring, episode)
    up to 4096 episodes, the VALU lockstep kernel above, the MFMA kernel from 12 episodes.  The fixtures and most
    cases in these files are small, so rerun both GRU files with each path forced for every size."""
    import subprocess, sys
    env = {**os.environ, **knobs}
    here = os.path.dirname(os.path.abspath(__file__))
    out = subprocess.run([sys.executable, "-m", "pytest", "-q", "-x", os.path.join(here, "test_gpu_gru.py"),
                          os.path.join(here, "test_gpu_lander.py"), "-k", "not every_kernel_path", "-m", "gpu"],
                         capture_output=True, text=True, env=env, timeout=1500, cwd=os.path.dirname(here))
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-2000:]


@pytest.mark.parametrize("E", [1, 3, 4, 5, 6, 8])
def test_gru_rollout_on_4x4x1_mfma_blocks_bit_exact(E):
    """The CartPole GRU rollout with the policy step on v_mfma_f32_4x4x1_16b_f32 (ses_gru_mfma4.h; knob gru_mfma4_min_e): up to 8
    episodes as two column blocks of four, one k per instruction -- the canonical chains bit for bit, so returns equal the C
    oracle's in both modes, masked and unmasked, for full and ragged column blocks."""
    from ses import HipES
    rng = np.random.RandomState(40 + E)
    n = 4200 // E + 1                                            # past the episode-parallel small-population path
    theta = (rng.randn(n, 6562) * 0.4).astype(np.float32)
    init = rng.uniform(-0.05, 0.05, (n, E, 4)).astype(np.float32)
    for pomdp, mask in ((True, 0b1010), (False, 0)):
        es = HipES("CartPole-v1", 4, 2, True, True, pomdp=pomdp, max_step=120, eval_ep_num=E)
        es.set_tuning("gru_mfma4_min_e", 1)
        o_fit, _, o_steps = co.rollout_cartpole(theta, init, E, 120, gru=True, obs_mask=mask)
        for mode in (0, 1):
            fit, _, ep_steps = es.rollout(dev(theta), dev(init), mode=mode, want_episodes=True)
            assert np.array_equal(ep_steps.cpu().numpy(), o_steps), (E, pomdp, mode)
            assert np.array_equal(bits(fit.cpu().numpy()), bits(o_fit))
        es.close()
    assert o_steps.max() > 50


@pytest.mark.parametrize("E", [1, 2, 3, 8, 9, 11])
def test_gru_lockstep_episode_batches(E):
    """The VALU lockstep kernel advances up to 8 episodes together: odd counts pad the last pair, 9 and 11 run a
    second batch (8 + 1, 8 + 3).  All must equal the oracle bit for bit."""
    from ses import HipES
    rng = np.random.RandomState(100 + E)
    n = 45
    theta = (rng.randn(n, 6562) * 0.4).astype(np.float32)
    init = rng.uniform(-0.05, 0.05, (n, E, 4)).astype(np.float32)
    es = HipES("CartPole-v1", 4, 2, True, True, pomdp=True, max_step=150, eval_ep_num=E)
    es.set_tuning("gru_mfma4_min_e", 0)                           # (7 and 8 episodes take the 4x4x1 MFMA step by default since round 6)
    o_fit, _, o_steps = co.rollout_cartpole(theta, init, E, 150, gru=True, obs_mask=0b1010)
    for mode in (0, 1):
        fit, _, ep_steps = es.rollout(dev(theta), dev(init), mode=mode, want_episodes=True)
        assert np.array_equal(ep_steps.cpu().numpy(), o_steps), (E, mode)
        assert np.array_equal(bits(fit.cpu().numpy()), bits(o_fit))
    es.close()
