"""ctypes front-end of the C oracle (oracle/ses_oracle.c).

TEST INFRASTRUCTURE.  Only tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg may import this module; the product path under
simple-es_amd/ never does.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libses_oracle.so")

MODE_EPISODIC = 0
MODE_FIXED_LENGTH = 1
HIDDEN = 32


def build(force=False):
    """Compile the C restatement with the recipe in oracle/Makefile."""
    world = os.path.join(os.path.dirname(_HERE), "simple-es_amd", "csrc")       # the Box2D-style world: one text, owned by the product
    srcs = [os.path.join(_HERE, f) for f in os.listdir(_HERE) if f.endswith((".c", ".cpp", ".h"))]
    srcs += [os.path.join(world, f) for f in ("ses_b2.h", "ses_b2_toi.h", "ses_b2_shapes.h", "ses_lander_env.h", "ses_walker_env.h")]
    src_m = max(os.path.getmtime(f) for f in srcs)

    def stale():
        # (both libraries of the Makefile's `all`: the oracle and its all-iterations checker build of the Box2D-style world)
        return any(not os.path.exists(so) or os.path.getmtime(so) < src_m
                   for so in (_SO, os.path.join(os.path.dirname(_SO), "libses_b2_allits.so")))

    if force or stale():
        # One builder at a time: the bench's CPU-baseline pool has a worker per host core, each of which gets here when the
        # library is older than a source (256 concurrent `make`s over one _build/ once left half-written objects, workers
        # that died loading them, and a Pool.map that waited for ever).  The others wait for the lock and find it built.
        import fcntl
        os.makedirs(os.path.dirname(_SO), exist_ok=True)
        with open(os.path.join(os.path.dirname(_SO), ".build.lock"), "w") as lock:
            fcntl.flock(lock, fcntl.LOCK_EX)
            try:
                if force or stale():
                    subprocess.check_call(["make", "-s", "-C", _HERE])
            finally:
                fcntl.flock(lock, fcntl.LOCK_UN)
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = ctypes.CDLL(build())
        _lib.o_param_count.restype = ctypes.c_int
    return _lib


def _p(a):
    return None if a is None else a.ctypes.data_as(ctypes.c_void_p)


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def param_count(S, A, gru):
    return lib().o_param_count(int(S), int(A), int(bool(gru)))


def philox_raw(ctr, key):
    ctr = np.ascontiguousarray(ctr, dtype=np.uint32)
    key = np.ascontiguousarray(key, dtype=np.uint32)
    out = np.empty(4, dtype=np.uint32)
    lib().o_philox_raw(_p(ctr), _p(key), _p(out))
    return out


def noise(seed, gen, first_row, n_rows, P):
    eps = np.empty((n_rows, P), dtype=np.float32)
    lib().o_noise(ctypes.c_uint64(seed), ctypes.c_uint64(gen), ctypes.c_int64(first_row),
                  ctypes.c_int(n_rows), ctypes.c_int(P), _p(eps))
    return eps


def perturb(parents, parent_idx, sigma, seed, gen, first_row, n_rows):
    parents = np.atleast_2d(_f32(parents))
    P = parents.shape[1]
    pidx = None if parent_idx is None else np.ascontiguousarray(parent_idx, dtype=np.int32)
    theta = np.empty((n_rows, P), dtype=np.float32)
    lib().o_perturb(_p(parents), _p(pidx), ctypes.c_float(sigma), ctypes.c_uint64(seed), ctypes.c_uint64(gen),
                    ctypes.c_int64(first_row), ctypes.c_int(n_rows), ctypes.c_int(P), _p(theta))
    return theta


def init_states_uniform(seed, gen, first_row, n_rows, E, S, shared, lo=-0.05, hi=0.05):
    out = np.empty((n_rows, E, S), dtype=np.float32)
    lib().o_init_states_uniform(ctypes.c_uint64(seed), ctypes.c_uint64(gen), ctypes.c_int64(first_row),
                                ctypes.c_int(n_rows), ctypes.c_int(E), ctypes.c_int(S), ctypes.c_int(int(shared)),
                                ctypes.c_float(lo), ctypes.c_float(hi), _p(out))
    return out


def policy_forward(S, A, discrete, gru, theta, obs, h=None):
    """n independent forwards.  Returns (action[n] i32, logits[n,A], act[n,A], h_next or None)."""
    theta = np.atleast_2d(_f32(theta))
    obs = np.atleast_2d(_f32(obs))
    n = obs.shape[0]
    assert theta.shape == (n, param_count(S, A, gru)), theta.shape
    logits = np.empty((n, A), dtype=np.float32)
    act = np.empty((n, A), dtype=np.float32)
    action = np.empty(n, dtype=np.int32)
    hh = None
    if gru:
        hh = np.zeros((n, HIDDEN), dtype=np.float32) if h is None else _f32(h).copy()
    lib().o_policy_forward(ctypes.c_int(S), ctypes.c_int(A), ctypes.c_int(int(discrete)), ctypes.c_int(int(gru)),
                           ctypes.c_int(n), _p(theta), _p(obs), _p(hh), _p(logits), _p(act), _p(action))
    return action, logits, act, hh


def cartpole_step_soa(mode, max_step, x, xd, th, thd, action, ret, status):
    """In-place SoA step (arrays must be contiguous f32 / i32 / u32)."""
    n = x.shape[0]
    lib().o_cartpole_step_soa(ctypes.c_int(n), ctypes.c_int(mode), ctypes.c_int(max_step),
                              _p(x), _p(xd), _p(th), _p(thd), _p(action), _p(ret), _p(status))


def rollout_cartpole(theta, init, E, max_step, *, S=4, A=2, discrete=True, gru=False,
                     mode=MODE_EPISODIC, obs_mask=0, physics64=False):
    """Returns (fitness[N] f32, ep_return[N,E] f64, ep_steps[N,E] i32).
    physics64: gym-order float64 dynamics instead of the folded-constant fp32 ones."""
    theta = np.atleast_2d(_f32(theta))
    N = theta.shape[0]
    init = _f32(init)
    per = 1 if init.ndim == 3 else 0
    assert init.shape[-2:] == (E, 4), init.shape
    if per:
        assert init.shape[0] == N
    ep_ret = np.empty((N, E), dtype=np.float64)
    ep_steps = np.empty((N, E), dtype=np.int32)
    fit = np.empty(N, dtype=np.float32)
    fn = lib().o_rollout_cartpole64 if physics64 else lib().o_rollout_cartpole
    fn(ctypes.c_int(S), ctypes.c_int(A), ctypes.c_int(int(discrete)), ctypes.c_int(int(gru)),
                             ctypes.c_int(N), ctypes.c_int(E), ctypes.c_int(max_step), ctypes.c_int(mode),
                             ctypes.c_uint32(obs_mask), _p(theta), _p(init), ctypes.c_int(per),
                             _p(ep_ret), _p(ep_steps), _p(fit))
    return fit, ep_ret, ep_steps


def spread_obs(n_agents, state, agent):
    state = _f32(state)
    obs = np.empty(6 * n_agents, dtype=np.float32)
    lib().o_spread_obs(ctypes.c_int(n_agents), _p(state), ctypes.c_int(agent), _p(obs))
    return obs


def spread_step(n_agents, state, action):
    """state float32[6n] updated in place; returns the team reward of the cycle."""
    action = np.ascontiguousarray(action, dtype=np.int32)
    fn = lib().o_spread_step
    fn.restype = ctypes.c_float
    return float(fn(ctypes.c_int(n_agents), _p(state), _p(action)))


def rollout_spread(theta, init, E, n_agents, max_cycles=25):
    """Returns (fitness[N] f32, ep_return[N,E] f64)."""
    theta = np.atleast_2d(_f32(theta))
    N = theta.shape[0]
    init = _f32(init)
    per = 1 if init.ndim == 3 else 0
    assert init.shape[-2:] == (E, 4 * n_agents), init.shape
    ep_ret = np.empty((N, E), dtype=np.float64)
    fit = np.empty(N, dtype=np.float32)
    lib().o_rollout_spread(ctypes.c_int(n_agents), ctypes.c_int(N), ctypes.c_int(E), ctypes.c_int(max_cycles),
                           _p(theta), _p(init), ctypes.c_int(per), _p(ep_ret), _p(fit))
    return fit, ep_ret


def rollout_lander(theta, init, E, max_step=300, *, gru=True, obs_mask=0b101100):
    """LunarLanderContinuous-v2 population rollout (ses_lander_env.h over the Box2D-style world of ses_b2.h).
    Returns (fitness[N], ep_return[N,E] f64, ep_steps[N,E])."""
    theta = np.atleast_2d(_f32(theta))
    N = theta.shape[0]
    init = _f32(init)
    per = 1 if init.ndim == 3 else 0
    assert init.shape[-2:] == (E, 16), init.shape
    ep_ret = np.empty((N, E), dtype=np.float64)
    ep_steps = np.empty((N, E), dtype=np.int32)
    fit = np.empty(N, dtype=np.float32)
    lib().o_rollout_lander(ctypes.c_int(int(gru)), ctypes.c_int(N), ctypes.c_int(E), ctypes.c_int(max_step),
                           ctypes.c_uint32(obs_mask), _p(theta), _p(init), ctypes.c_int(per), _p(ep_ret), _p(ep_steps),
                           _p(fit))
    return fit, ep_ret, ep_steps


class LanderSim:
    """one LunarLanderContinuous-v2 env driven step by step (used by oracle/lander_env.py)"""

    def __init__(self):
        self._buf = ctypes.create_string_buffer(lib().o_lander_state_size())
        lib().o_lander_step.restype = ctypes.c_float

    def debug(self):
        """bodies[3,6] = (cx, cy, angle, vx, vy, omega) of hull / leg -1 / leg +1 and a dict of solver facts"""
        bodies = np.empty((3, 6), dtype=np.float32)
        ints = np.zeros(8, dtype=np.int32)
        lib().o_lander_debug(self._buf, _p(bodies), _p(ints))
        keys = ("unused", "limit0", "limit1", "game_over", "awake", "leg0", "leg1", "contact_points")
        return bodies, dict(zip(keys, ints.tolist()))

    def reset(self, u16):
        u16 = _f32(u16)
        obs = np.empty(8, dtype=np.float32)
        lib().o_lander_reset(self._buf, _p(u16), _p(obs))
        return obs

    def step(self, a0, a1):
        obs = np.empty(8, dtype=np.float32)
        done = ctypes.c_int32(0)
        r = lib().o_lander_step(self._buf, ctypes.c_float(a0), ctypes.c_float(a1), _p(obs), ctypes.byref(done))
        return obs, float(r), bool(done.value)


def rollout_walker(theta, init, E, max_step=300, *, gru=False):
    """BipedalWalker-v3 population rollout (ses_walker_env.h over the Box2D-style world of ses_b2.h).
    init: [E, 4] or [N, E, 4] rows (force uniform, terrain key words, pad).  Returns (fitness, ep_return f64, ep_steps)."""
    theta = np.atleast_2d(_f32(theta))
    N = theta.shape[0]
    init = _f32(init)
    per = 1 if init.ndim == 3 else 0
    assert init.shape[-2:] == (E, 4), init.shape
    ep_ret = np.empty((N, E), dtype=np.float64)
    ep_steps = np.empty((N, E), dtype=np.int32)
    fit = np.empty(N, dtype=np.float32)
    lib().o_rollout_walker(ctypes.c_int(int(gru)), ctypes.c_int(N), ctypes.c_int(E), ctypes.c_int(max_step),
                           _p(theta), _p(init), ctypes.c_int(per), _p(ep_ret), _p(ep_steps), _p(fit))
    return fit, ep_ret, ep_steps


def toi_probe(body, edge, start, end):
    """b2TimeOfImpact of lander body `body`'s polygon swept from start = (cx, cy, angle) to end against edge = (x1, y1, x2,
    y2): (state, t) with state 0 failed / 1 overlapped / 2 touching / 3 separated"""
    t = ctypes.c_float(0.0)
    e, a, b = _f32(edge), _f32(start), _f32(end)
    state = lib().o_toi_probe(int(body), _p(e), _p(a), _p(b), ctypes.byref(t))
    return int(state), float(t.value)


def walker_body_props():
    """[5, 5] float32: mass, inertia about the centre of mass, local centre x, y, mixed friction -- the float32 world's tables"""
    props = np.empty((5, 5), dtype=np.float32)
    lib().o_walker_body_props(_p(props))
    return props


class WalkerSim:
    """one BipedalWalker-v3 env driven step by step"""

    def __init__(self):
        self._buf = ctypes.create_string_buffer(lib().o_walker_state_size())
        lib().o_walker_step.restype = ctypes.c_float

    def reset(self, u4):
        u4 = _f32(u4)
        obs = np.empty(24, dtype=np.float32)
        lib().o_walker_reset(self._buf, _p(u4), _p(obs))
        return obs

    def step(self, action):
        a = _f32(np.asarray(action).reshape(4))
        obs = np.empty(24, dtype=np.float32)
        done = ctypes.c_int32(0)
        r = lib().o_walker_step(self._buf, _p(a), _p(obs), ctypes.byref(done))
        return obs, float(r), bool(done.value)

    def debug(self):
        bodies = np.empty((5, 6), dtype=np.float32)
        terrain = np.empty(200, dtype=np.float32)
        ints = np.zeros(10, dtype=np.int32)
        lib().o_walker_debug(self._buf, _p(bodies), _p(terrain), _p(ints))
        return bodies, terrain, {"game_over": int(ints[0]), "contact_points": int(ints[1]), "limits": ints[2:6].tolist(),
                                 "manifolds_per_leg": ints[6:10].tolist()}
