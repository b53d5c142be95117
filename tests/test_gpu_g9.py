"""Fixture G9 on the HIP path: policies that live long (trained checkpoints + perturbations; returns, episode lengths
and GRU trajectories from the imported reference, tests/golden/make_golden.py g9).

Bars: device == C oracle BIT FOR BIT (fitness, per-episode returns, episode lengths) in both rollout modes and for every
lanes-per-env variant of the MLP kernel; device vs the reference: CartPole returns within 1e-4 (north_star) -- at these
horizons that means no argmax flipped in 633 000 env steps -- and LunarLander as tests/test_oracle_g9.py states it
(episode lengths equal; returns inside the reference's own one-ulp sensitivity)."""
import os

import numpy as np
import pytest
import torch

from oracle import c_oracle as co

pytestmark = pytest.mark.gpu
RETURN_TOL = 1e-4


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def host(t):
    return t.detach().cpu().numpy()


def bits(a):
    a = np.ascontiguousarray(a)
    return a.view(np.uint32 if a.dtype.itemsize == 4 else np.uint64)


@pytest.fixture(scope="module")
def g9(golden_dir):
    return np.load(os.path.join(golden_dir, "g9_long.npz"))


@pytest.mark.parametrize("lpe", [0, 1, 4, 16, 32])
def test_g9_cartpole_mlp(g9, lpe):
    from ses import HipES
    theta, init = g9["mlp_theta"], g9["init_states"]
    o_fit, o_ret, o_steps = co.rollout_cartpole(theta, init, 5, 500)
    es = HipES("CartPole-v1", 4, 2, True, False, max_step=500, eval_ep_num=5, lanes_per_env=lpe)
    for mode, packed in [(m, k) for m in (0, 1) for k in ((0, 1) if lpe in (0, 16) else (0,))]:
        es.set_tuning("rollout_packed", packed)                      # the scalar step and the packed step of lone waves
        fit, ep_ret, ep_steps = es.rollout(dev(theta), dev(init), mode=mode, want_episodes=True)
        assert np.array_equal(host(ep_steps), o_steps), f"mode {mode}"
        assert np.array_equal(bits(host(fit)), bits(o_fit)) and np.array_equal(host(ep_ret), o_ret)
        assert np.array_equal(host(ep_steps), g9["mlp_steps"]), "an episode ends at a different step than the reference's"
        assert np.abs(host(fit).astype(np.float64) - g9["mlp_returns"]).max() <= RETURN_TOL
    es.close()


def test_g9_pomdp_cartpole_gru(g9):
    from ses import HipES
    theta, init = g9["gru_theta"], g9["init_states"]
    o_fit, _, o_steps = co.rollout_cartpole(theta, init, 5, 500, gru=True, obs_mask=0b1010)
    es = HipES("CartPole-v1", 4, 2, True, True, pomdp=True, max_step=500, eval_ep_num=5)
    for mode in (0, 1):
        fit, _, ep_steps = es.rollout(dev(theta), dev(init), mode=mode, want_episodes=True)
        assert np.array_equal(host(ep_steps), o_steps) and np.array_equal(bits(host(fit)), bits(o_fit))
        assert np.array_equal(host(ep_steps), g9["gru_steps"])
        assert np.abs(host(fit).astype(np.float64) - g9["gru_returns"]).max() <= RETURN_TOL
    es.close()


@pytest.mark.parametrize("tag,S,A,disc", [("gru", 4, 2, True), ("lander", 8, 4, False)])
def test_g9_gru_trajectories_on_device(g9, tag, S, A, disc):
    """The device forward along whole episodes of the reference module: teacher-forced against the reference's hidden
    state (tolerance of G1), free-running against the C oracle bit for bit."""
    from ses import HipES
    h = HipES(None, S, A, disc, True)
    theta, obs = g9[f"traj_{tag}_theta"], g9[f"traj_{tag}_obs"]
    H, act = g9[f"traj_{tag}_h"], g9[f"traj_{tag}_act"]
    n, T, _ = obs.shape
    d_theta = dev(theta)
    free_d, free_o = dev(np.zeros((n, 32), np.float32)), np.zeros((n, 32), np.float32)
    for t in range(T):
        hid = dev(H[:, t - 1] if t else np.zeros((n, 32), np.float32))
        action, logits, a_out = h.policy_forward(d_theta, dev(obs[:, t]), hid)
        np.testing.assert_allclose(host(hid), H[:, t], rtol=0, atol=1e-5, err_msg=f"step {t}")
        if disc:
            assert np.array_equal(host(action), act[:, t, 0].astype(np.int32)), f"step {t}"
        else:
            np.testing.assert_allclose(host(a_out), act[:, t], rtol=0, atol=1e-5)
        f_action, f_logits, _ = h.policy_forward(d_theta, dev(obs[:, t]), free_d)
        _, o_logits, _, free_o = co.policy_forward(S, A, disc, True, theta, obs[:, t], free_o)
        assert np.array_equal(bits(host(free_d)), bits(free_o)) and np.array_equal(bits(host(f_logits)), bits(o_logits)), f"step {t}"
        if disc:
            assert np.array_equal(host(f_action), act[:, t, 0].astype(np.int32)), f"free-running, step {t}"
    h.close()


def test_g9_lander_gru_long_episodes(g9):
    from ses import HipES
    theta, init = g9["lander_theta"], g9["lander_init"]
    es = HipES("LunarLanderContinuous-v2", 8, 4, False, True, pomdp=True, max_step=300, eval_ep_num=3)
    fit, ep_ret, ep_steps = es.rollout(dev(theta), dev(init), want_episodes=True)
    o_fit, o_ret, o_steps = co.rollout_lander(theta, init, 3, 300)
    assert np.array_equal(host(ep_steps), o_steps)
    assert np.array_equal(bits(host(ep_ret)), bits(o_ret)) and np.array_equal(bits(host(fit)), bits(o_fit))
    assert np.array_equal(host(ep_steps), g9["lander_steps"])
    ref = g9["lander_returns"]
    dev_hip = np.abs(host(fit).astype(np.float64) - ref)
    dev_ref = np.abs(g9["lander_returns_ulp"] - ref)
    short = g9["lander_steps"].max(axis=1) < 120
    np.testing.assert_allclose(host(fit)[short].astype(np.float64), ref[short], rtol=1e-5, atol=1e-4)
    assert np.median(dev_hip) <= 1.5 * np.median(dev_ref) and dev_hip.max() <= dev_ref.max()
    es.close()
