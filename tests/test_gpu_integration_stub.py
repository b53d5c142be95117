"""INTEGRATION.md section 2 shows the binding a maintainer of the reference would add (a ctypes stub that replaces
`p.map(RolloutWorker, arguments)`, loop.py:66-79).  This test EXECUTES that stub as written in the document -- the first
python block of the section, taken from the file -- against module objects with the reference's `get_param_list()`
protocol, and checks what it returns against the oracle."""
import os
import re

import numpy as np
import pytest

from oracle import c_oracle as co

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_the_documented_ctypes_stub_runs_and_matches_the_oracle(monkeypatch):
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    section = text[text.index("## 2. Binding the C ABI"):text.index("## 3.")]
    code = re.search(r"```python\n(.*?)```", section, flags=re.S).group(1)
    assert "class HipRollout" in code and "ses_rollout" in code
    monkeypatch.chdir(ROOT)                                    # the stub loads "simple-es_amd/libses_hip.so"
    ns = {}
    exec(compile(code, "INTEGRATION.md#2", "exec"), ns)
    from networks.neural_network import GymEnvModel
    rng = np.random.RandomState(0)
    offsprings = []
    for _ in range(9):
        m = GymEnvModel(4, 2, True, False)
        m.load_flat((rng.randn(m.param_count()) * 0.5).astype(np.float32))
        offsprings.append({"0": m})
    env_cfg = {"name": "CartPole-v1", "max_step": 500, "pomdp": False}
    net_cfg = {"num_state": 4, "num_action": 2, "discrete_action": True, "gru": False}
    roll = ns["HipRollout"](env_cfg, net_cfg, 5)
    for gen in range(2):                                       # two "generations": the stub advances its reset key
        results = roll(offsprings)
        assert isinstance(results, list) and len(results) == 9 and all(isinstance(r, float) for r in results)
        theta = np.stack([o["0"].flat() for o in offsprings])
        init = co.init_states_uniform(0, gen, 0, 9, 5, 4, False)
        want, _, _ = co.rollout_cartpole(theta, init, 5, 500)
        assert results == want.astype(np.float64).tolist()
    assert max(results) > min(results)
