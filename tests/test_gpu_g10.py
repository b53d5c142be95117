"""Fixture G10 on the HIP path: the MLP policies of conf/lunarlander.yaml and conf/bipedalwalker.yaml (first-generation and
trained), reference RolloutWorker returns over the build's env objects.  Device == host build of the world bit for bit (fitness,
per-episode returns, episode lengths); device vs the reference as tests/test_oracle_g10.py states it: inside the reference's own
one-ulp envelope, first-generation landers exactly, the 24-input forward step by step along a walker episode."""
import os

import numpy as np
import pytest
import torch

from oracle import c_oracle as co

pytestmark = pytest.mark.gpu


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def bits(a):
    a = np.ascontiguousarray(a)
    return a.view(np.uint32 if a.dtype.itemsize == 4 else np.uint64)


@pytest.fixture(scope="module")
def g10(golden_dir):
    return np.load(os.path.join(golden_dir, "g10_box2d_mlp.npz"))


@pytest.mark.parametrize("tag,env,S", [("lander", "LunarLanderContinuous-v2", 8), ("walker", "BipedalWalker-v3", 24)])
def test_g10_box2d_mlp_rollouts(g10, tag, env, S):
    from ses import HipES
    theta, init = g10[f"{tag}_theta"], g10[f"{tag}_init"]
    es = HipES(env, S, 4, False, False, pomdp=False, max_step=300, eval_ep_num=3)
    fit, ep_ret, ep_steps = es.rollout(dev(theta), dev(init), want_episodes=True)
    if tag == "lander":
        o_fit, o_ret, o_steps = co.rollout_lander(theta, init, 3, 300, gru=False, obs_mask=0)
    else:
        o_fit, o_ret, o_steps = co.rollout_walker(theta, init, 3, 300)
    fit, ep_ret, ep_steps = fit.cpu().numpy(), ep_ret.cpu().numpy(), ep_steps.cpu().numpy()
    assert np.array_equal(ep_steps, o_steps)
    assert np.array_equal(bits(ep_ret), bits(o_ret)) and np.array_equal(bits(fit), bits(o_fit))
    ref, ref_steps = g10[f"{tag}_returns"], g10[f"{tag}_steps"]
    d, d_ref = np.abs(fit.astype(np.float64) - ref), np.abs(g10[f"{tag}_returns_ulp"] - ref)
    assert np.median(d) <= 1.5 * np.median(d_ref) + 1e-3 and d.max() <= d_ref.max()
    assert np.mean(ep_steps == ref_steps) >= np.mean(g10[f"{tag}_steps_ulp"] == ref_steps) - 0.1
    if tag == "lander":
        assert np.array_equal(ep_steps[:12], ref_steps[:12])
        np.testing.assert_allclose(fit[:12].astype(np.float64), ref[:12], rtol=1e-5, atol=1e-3)
    es.close()


def test_g10_forward_24_inputs_on_device(g10):
    from ses import HipES
    obs, ref_logits, ref_act = g10["fwd_walker_obs"], g10["fwd_walker_logits"], g10["fwd_walker_act"]
    h = HipES(None, 24, 4, False, False)
    theta = np.repeat(g10["fwd_walker_theta"][None], len(obs), axis=0)
    _, logits, act = h.policy_forward(dev(theta), dev(obs))
    _, o_logits, o_act, _ = co.policy_forward(24, 4, False, False, theta, obs)
    assert np.array_equal(bits(logits.cpu().numpy()), bits(o_logits)) and np.array_equal(bits(act.cpu().numpy()), bits(o_act))
    w2 = g10["fwd_walker_theta"][24 * 32 + 32:24 * 32 + 32 + 4 * 32].reshape(4, 32)
    bound = 2e-6 * np.abs(w2).sum(axis=1)[None, :] + 3e-6 * np.abs(ref_logits)
    assert (np.abs(logits.cpu().numpy() - ref_logits) <= bound).all()
    assert np.abs(act.cpu().numpy() - ref_act).max() < 2e-4
    h.close()
