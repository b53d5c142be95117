"""BipedalWalker-v3 (csrc/ses_walker.h: gym's env on the Box2D-style world of ses_b2.h) on the HIP path: MLP rollouts
bit-exact against the CPU build of the same world, and the reference's conf/bipedalwalker.yaml end to end."""
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


@pytest.mark.parametrize("n,E,T,lpe,epw", [(1, 1, 40, 0, 0), (37, 3, 150, 0, 0), (37, 3, 150, 8, 0), (37, 3, 150, 4, 0), (37, 3, 150, 2, 0),
                                           (37, 3, 150, 1, 0), (37, 3, 150, 32, 0), (37, 3, 150, 2, 20), (37, 3, 150, 0, 7), (37, 3, 150, 4, 3),
                                           (37, 3, 150, 1, 64), (37, 3, 150, 16, 9)])
def test_walker_population_bit_exact(n, E, T, lpe, epw):
    """lpe: lanes per env of the rollout kernel (0 = the library's choice); 1 and 2 stream the weights.  epw: different envs
    per wave (0 = 64 / lpe or, with lpe = 0, as few as the wave slots allow; the lane groups past the last env shadow it;
    values beyond 64 / lpe are clamped)."""
    from ses import HipES
    rng = np.random.RandomState(n)
    es = HipES("BipedalWalker-v3", 24, 4, False, False, max_step=T, eval_ep_num=E)
    es.set_tuning("box2d_lanes_per_env", lpe)
    es.set_tuning("box2d_envs_per_wave", epw)
    assert es.P == 932
    theta = (rng.randn(n, es.P) * rng.choice([0.1, 0.5, 2.0], size=(n, 1))).astype(np.float32)
    init = es.init_states_uniform(3, 1, 20, n)                          # [n, E, 4] uniforms in [0,1)
    want_init = co.init_states_uniform(3, 1, 20, n, E, 4, False, 0.0, 1.0)
    assert np.array_equal(init.cpu().numpy().view(np.uint32), want_init.view(np.uint32))
    fit, ep_ret, ep_steps = es.rollout(dev(theta), init, want_episodes=True)
    o_fit, o_ret, o_steps = co.rollout_walker(theta, want_init, E, T)
    assert np.array_equal(ep_steps.cpu().numpy(), o_steps)
    assert np.array_equal(ep_ret.cpu().numpy().view(np.uint64), o_ret.view(np.uint64))
    assert np.array_equal(fit.cpu().numpy().view(np.uint32), o_fit.view(np.uint32))
    if n > 1:
        assert o_steps.min() < T and o_ret.min() < -90                  # falls happen in this population
    es.close()


def test_walker_needs_the_mlp_policy():
    from ses import HipES, SesError
    with pytest.raises(SesError):
        HipES("BipedalWalker-v3", 24, 4, False, True, max_step=10, eval_ep_num=1)


def test_bipedalwalker_yaml_runs_and_improves(tmp_path, monkeypatch):
    import builder
    monkeypatch.chdir(tmp_path)
    cfg = yaml.load(open(os.path.join(SRC, "conf", "bipedalwalker.yaml")), Loader=yaml.FullLoader)
    loop = builder.build_loop(cfg, 12, 1, 2, False, 10 ** 9)
    with contextlib.redirect_stdout(io.StringIO()):
        loop.run()
    best = [b for b, _ in loop.history]
    assert len(best) == 12 and np.isfinite(best).all()
    assert max(best[-4:]) >= best[0] - 1e-6, best                      # the elites are kept: the best never gets worse
