"""The oracle's Philox4x32-10 against the published Random123 known-answer vectors
(Salmon, Moraes, Dror, Shaw, SC'11; kat_vectors of the Random123 distribution), plus sanity
statistics of the Box-Muller normals built on it.  rocRAND's device engine, which the HIP
perturbation kernel uses, implements the same function (checked on the GPU in test_gpu_parity.py)."""
import numpy as np
from scipy import stats

from oracle import c_oracle as co

KAT = [
    ([0, 0, 0, 0], [0, 0], [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]),
    ([0xffffffff] * 4, [0xffffffff] * 2, [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]),
    ([0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344], [0xa4093822, 0x299f31d0],
     [0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1]),
]


def test_philox_known_answers():
    for ctr, key, want in KAT:
        assert list(co.philox_raw(ctr, key)) == want


def test_normals_statistics():
    eps = co.noise(seed=0, gen=0, first_row=0, n_rows=2048, P=226)
    assert abs(eps.mean()) < 5e-3 and abs(eps.std() - 1) < 5e-3
    assert stats.kstest(eps.ravel()[:200_000], "norm").pvalue > 1e-3
    assert np.isfinite(eps).all() and np.abs(eps).max() < 6.8       # 32-bit uniform -> |z| <= 6.76


def test_noise_is_counter_based():
    """Row i depends only on (seed, gen, global row, column): shards reproduce the full matrix."""
    full = co.noise(5, 3, 0, 64, 226)
    part = co.noise(5, 3, 40, 24, 226)
    assert np.array_equal(full[40:], part)
    assert not np.array_equal(full, co.noise(5, 4, 0, 64, 226))
    assert not np.array_equal(full, co.noise(6, 3, 0, 64, 226))
    # P not a multiple of 4: the tail of the last Philox quad is dropped, nothing shifts
    assert np.array_equal(co.noise(5, 3, 0, 8, 225), full[:8, :225])


def test_perturb_parent_map():
    parents = np.random.RandomState(0).randn(3, 226).astype(np.float32)
    idx = np.array([-1, -2, 0, 1, 2, 2], np.int32)
    th = co.perturb(parents, idx, 0.5, 1, 2, 10, 6)
    assert np.array_equal(th[0], parents[0]) and np.array_equal(th[1], parents[1])
    eps = co.noise(1, 2, 10, 6, 226)
    for i, k in ((2, 0), (3, 1), (4, 2), (5, 2)):
        want = (parents[k].astype(np.float64) + 0.5 * eps[i].astype(np.float64)).astype(np.float32)  # exact fma
        assert np.array_equal(th[i], want)


def test_init_states_range():
    s = co.init_states_uniform(0, 0, 0, 64, 5, 4, shared=False)
    assert s.min() >= -0.05 and s.max() <= 0.05 and abs(s.mean()) < 2e-3
    sh = co.init_states_uniform(0, 0, 7, 3, 5, 4, shared=True)
    assert np.array_equal(sh[0], sh[1]) and np.array_equal(sh[1], sh[2])
