"""LunarLanderContinuous-v2 (csrc/ses_lander.h: gym's env on the Box2D-style world of ses_b2.h) on the HIP path: GRU
(wave per offspring / per episode) and MLP (8 lanes per env) rollouts, continuous tanh head, float rewards -- bit-exact
against the CPU build of the same world, fixture G8 within float tolerance, and the reference's
conf/lunarlander_openai.yaml end to end."""
import contextlib
import io
import os

import numpy as np
import pytest
import torch
import yaml

from oracle import c_oracle as co

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "simple-es_amd")


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def test_lander_gru_golden_and_oracle(golden_dir):
    from ses import HipES
    g = np.load(os.path.join(golden_dir, "g8_lander.npz"))
    es = HipES("LunarLanderContinuous-v2", 8, 4, False, True, pomdp=True, max_step=300, eval_ep_num=3)
    fit, ep_ret, ep_steps = es.rollout(dev(g["theta"]), dev(g["init"]), want_episodes=True)
    o_fit, o_ret, o_steps = co.rollout_lander(g["theta"], g["init"], 3, 300)
    assert np.array_equal(ep_steps.cpu().numpy(), o_steps)
    assert np.array_equal(ep_ret.cpu().numpy().view(np.uint64), o_ret.view(np.uint64)), "episode returns differ from the oracle"
    assert np.array_equal(fit.cpu().numpy().view(np.uint32), o_fit.view(np.uint32))
    np.testing.assert_allclose(fit.cpu().numpy().astype(np.float64), g["returns"], rtol=1e-5, atol=1e-4)   # observed: 7.8e-6 relative
    es.close()


@pytest.mark.parametrize("gru,pomdp,lpe", [(True, False, 0), (False, True, 0), (False, False, 0), (False, False, 4),
                                           (False, True, 2), (False, False, 1), (False, False, 8), (False, True, 16), (False, False, 64)])
def test_lander_population_bit_exact(gru, pomdp, lpe):
    """lpe: lanes per env of the MLP rollout kernel (0 = the library's choice, 8 at this size)."""
    from ses import HipES
    rng = np.random.RandomState(int(gru) * 2 + int(pomdp))
    n = 70
    es = HipES("LunarLanderContinuous-v2", 8, 4, False, gru, pomdp=pomdp, max_step=250, eval_ep_num=4)
    es.set_tuning("box2d_lanes_per_env", lpe)
    theta = (rng.randn(n, es.P) * rng.choice([0.05, 0.3, 1.0], size=(n, 1))).astype(np.float32)
    init = es.init_states_uniform(11, 2, 50, n)                         # [n, 4, 16] uniforms in [0,1)
    want_init = co.init_states_uniform(11, 2, 50, n, 4, 16, False, 0.0, 1.0)
    assert np.array_equal(init.cpu().numpy().view(np.uint32), want_init.view(np.uint32))
    fit, ep_ret, ep_steps = es.rollout(dev(theta), init, want_episodes=True)
    o_fit, o_ret, o_steps = co.rollout_lander(theta, want_init, 4, 250, gru=gru, obs_mask=0b101100 if pomdp else 0)
    assert np.array_equal(ep_steps.cpu().numpy(), o_steps)
    assert np.array_equal(ep_ret.cpu().numpy().view(np.uint64), o_ret.view(np.uint64))
    assert np.array_equal(fit.cpu().numpy().view(np.uint32), o_fit.view(np.uint32))
    assert o_steps.min() < 250 and o_ret.min() < -100                    # crashes happen in this population
    es.close()


def test_lunarlander_openai_yaml_improves(tmp_path, monkeypatch):
    import builder
    monkeypatch.chdir(tmp_path)
    cfg = yaml.load(open(os.path.join(SRC, "conf", "lunarlander_openai.yaml")), Loader=yaml.FullLoader)
    cfg["strategy"]["offspring_num"] = 512
    loop = builder.build_loop(cfg, 60, 1, 5, False, 10 ** 9)
    with contextlib.redirect_stdout(io.StringIO()):
        loop.run()
    best = [b for b, _ in loop.history]
    assert np.mean(best[-10:]) > np.mean(best[:5]) + 50, best[::5]
