"""End-to-end learning checks on the HIP path.

The reference publishes exactly one learning result (README.md:42): on POMDP CartPole (velocities zeroed,
envs/gym_wrapper.py:69-77) the GRU policy trained with simple_evolution reaches the maximum score of 500 while
the MLP ("ANN") stays around 60.  The drop-in path must reproduce that qualitative behaviour."""
import contextlib
import io
import os

import pytest
import yaml

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "simple-es_amd")


def run(gru, gens, tmp_path, monkeypatch):
    import builder
    monkeypatch.chdir(tmp_path)
    cfg = yaml.load(open(os.path.join(SRC, "conf", "cartpole_pomdp_gru.yaml")), Loader=yaml.FullLoader)
    cfg["network"]["gru"] = gru
    loop = builder.build_loop(cfg, gens, 1, 5, False, 10 ** 9)
    with contextlib.redirect_stdout(io.StringIO()):
        loop.run()
    return [b for b, _ in loop.history]


def test_pomdp_cartpole_gru_reaches_500_mlp_does_not(tmp_path, monkeypatch):
    gru_best = run(True, 150, tmp_path, monkeypatch)
    mlp_best = run(False, 150, tmp_path, monkeypatch)
    assert max(gru_best) == 500.0 and min(gru_best[-20:]) >= 450, gru_best[::10]
    assert max(mlp_best) < 120, mlp_best[::10]          # README: "about 60"
