"""Python face of oracle/lander64.c -- the independently written float64 LunarLander (TEST INFRASTRUCTURE).

The integration shares no code with the product's Box2D-style world; what it shares are the INPUTS: the 16 reset uniforms
and, per step, the two engine-dispersion numbers, which the device draws from Philox keyed by the episode (u[14], u[15])
and the step counter -- reproduced here through the oracle's raw Philox so that both integrations see the same noise."""
import ctypes

import numpy as np

from . import c_oracle


def dispersion(key0, key1, step):
    """The two uniforms in (-1, 1) of env step `step` (csrc/ses_lander.h B2_DISPERSION), as float32 values."""
    r = c_oracle.philox_raw(np.array([step, 0, 0, 2 << 24], np.uint32), np.array([key0, key1], np.uint32))
    unit = np.float32(r[:2].astype(np.float32)) * np.float32(2.0 ** -32) + np.float32(2.0 ** -33)
    d = unit.astype(np.float32) * np.float32(2.0) - np.float32(1.0)
    return float(d[0]), float(d[1])


class Lander64:
    def __init__(self):
        lib = c_oracle.lib()
        lib.l64_step.restype = ctypes.c_double
        self._lib = lib
        self._buf = ctypes.create_string_buffer(lib.l64_state_size())
        self._key = (0, 0)
        self._step = 0

    def reset(self, u16):
        u = np.ascontiguousarray(u16, dtype=np.float32)
        self._key = tuple(int(k) for k in u[14:16].view(np.uint32))
        self._step = 1                         # the reset's own no-op step consumed draw 0 (no engine fires: its values are unused)
        self._lib.l64_reset(self._buf, u.ctypes.data_as(ctypes.c_void_p))
        return self.obs()

    def adopt(self, bodies):
        """Start from the configuration another integration reached after its reset (c_oracle.LanderSim.debug()[0]): see
        l64_adopt in lander64.c for why the comparison starts after gym's leg snap."""
        b = np.ascontiguousarray(bodies, dtype=np.float64).reshape(3, 6)
        self._lib.l64_adopt(self._buf, b.ctypes.data_as(ctypes.c_void_p))
        return self.obs()

    def obs(self):
        o = np.empty(8, np.float64)
        self._lib.l64_obs(self._buf, o.ctypes.data_as(ctypes.c_void_p))
        return o

    def step(self, a0, a1):
        d0, d1 = dispersion(self._key[0], self._key[1], self._step)
        self._step += 1
        done = ctypes.c_int32(0)
        r = self._lib.l64_step(self._buf, ctypes.c_double(a0), ctypes.c_double(a1), ctypes.c_double(d0), ctypes.c_double(d1),
                               ctypes.byref(done))
        return self.obs(), float(r), bool(done.value)

    def debug(self):
        bodies = np.empty((3, 6), np.float64)
        flags = np.zeros(4, np.int32)
        self._lib.l64_debug(self._buf, bodies.ctypes.data_as(ctypes.c_void_p), flags.ctypes.data_as(ctypes.c_void_p))
        return bodies, dict(zip(("game_over", "awake", "leg0", "leg1"), flags.tolist()))
