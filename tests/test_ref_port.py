"""The reference-structured CPU port (oracle/ref_port.py: mp.Pool + batch-1 torch forward + Python env object) is
what bench.py times as `cpu_baseline`; its per-offspring returns are pinned to fixture G5 (returns of the
reference's own RolloutWorker)."""
import os

import numpy as np

from oracle import ref_port


def test_port_returns_match_reference_fixture(golden_dir):
    g = np.load(os.path.join(golden_dir, "g56_rollouts.npz"))
    theta = g["g5_theta"][150:182]                                   # includes trained (long) policies
    rets, steps, _ = ref_port.run_generation(theta, g["init_states"], 5, 500, process_num=2)
    assert np.abs(rets - g["g5_returns"][150:182]).max() <= 1e-4
    assert steps == int(round(rets.sum() * 5))
    serial, _, _ = ref_port.run_generation(theta[:4], g["init_states"], 5, 500, process_num=1)
    assert np.array_equal(serial, rets[:4])                          # Pool.map preserves order and values


def test_fixed_length_variant_counts_every_step():
    theta = (np.random.RandomState(0).standard_normal((3, 226)) * 0.1).astype(np.float32)
    init = np.random.RandomState(0).uniform(-0.05, 0.05, (2, 4)).astype(np.float32)
    rets, steps, _ = ref_port.run_generation(theta, init, 2, 50, process_num=1, fixed_length=True)
    assert steps == 3 * 2 * 50
