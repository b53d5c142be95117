"""Python face of oracle/walker64.c -- the independently written float64 BipedalWalker (TEST INFRASTRUCTURE).

The integration shares no code with the product's Box2D-style world; what it shares are the INPUTS: the episode's 200
terrain heights, the actions and (through adopt) the configuration the float32 world is in after gym's reset."""
import ctypes

import numpy as np

from . import c_oracle


class Walker64:
    def __init__(self):
        lib = c_oracle.lib()
        lib.w64_step.restype = ctypes.c_double
        self._lib = lib
        self._buf = ctypes.create_string_buffer(lib.w64_state_size())

    def reset(self, terrain, force_u=0.5):
        t = np.ascontiguousarray(terrain, dtype=np.float64)
        assert t.shape == (200,)
        self._lib.w64_reset(self._buf, t.ctypes.data_as(ctypes.c_void_p), ctypes.c_double(float(force_u)))
        return self.obs()

    def adopt(self, bodies):
        """Start from the configuration another integration reached after its reset (c_oracle.WalkerSim.debug()[0]): see
        w64_adopt in walker64.c for why the comparison starts after gym's leg snap."""
        b = np.ascontiguousarray(bodies, dtype=np.float64).reshape(5, 6)
        self._lib.w64_adopt(self._buf, b.ctypes.data_as(ctypes.c_void_p))
        return self.obs()

    def obs(self):
        o = np.empty(24, np.float64)
        self._lib.w64_obs(self._buf, o.ctypes.data_as(ctypes.c_void_p))
        return o

    def step(self, action):
        a = np.ascontiguousarray(np.asarray(action, dtype=np.float64).reshape(4))
        done = ctypes.c_int32(0)
        r = self._lib.w64_step(self._buf, a.ctypes.data_as(ctypes.c_void_p), ctypes.byref(done))
        return self.obs(), float(r), bool(done.value)

    def debug(self):
        bodies = np.empty((5, 6), np.float64)
        flags = np.zeros(6, np.int32)
        props = np.empty((5, 4), np.float64)
        self._lib.w64_debug(self._buf, bodies.ctypes.data_as(ctypes.c_void_p), flags.ctypes.data_as(ctypes.c_void_p),
                            props.ctypes.data_as(ctypes.c_void_p))
        return bodies, {"game_over": int(flags[0]), "contact": flags[1:].tolist()}, props
