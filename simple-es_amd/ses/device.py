"""HipES: shape/dtype-checked torch-tensor front end of the C ABI (include/ses.h).

PyTorch-ROCm tensors are only the parameter/fitness containers; every compute step is a
hand-written gfx950 kernel inside libses_hip.so, enqueued on torch's current HIP stream.
All checks happen on the host BEFORE a kernel is launched: a wrong shape raises here, it never
reaches the GPU.
"""
import contextlib
import ctypes
import os
import sys

import torch

from . import _lib
from ._lib import (ENV_BIPEDALWALKER, ENV_CARTPOLE, ENV_LUNARLANDER, ENV_NONE, ENV_SIMPLE_SPREAD, HIDDEN, MODE_EPISODIC,
                   SesConfig, SesError, check)

ENV_IDS = {"CartPole-v1": ENV_CARTPOLE, "CartPole-v0": ENV_CARTPOLE, "simple_spread": ENV_SIMPLE_SPREAD,
           "LunarLanderContinuous-v2": ENV_LUNARLANDER, "BipedalWalker-v3": ENV_BIPEDALWALKER, None: ENV_NONE}


def param_count(num_state, num_action, gru):
    return _lib.load().ses_param_count(int(num_state), int(num_action), int(bool(gru)))


def _ptr(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


@contextlib.contextmanager
def _stdout_to_stderr():
    """fd-level redirect of stdout to stderr around a native call that prints (RCCL's banner): programs that emit
    machine-readable lines on stdout (bench.py, run_es.py --log pipelines) stay clean."""
    sys.stdout.flush()
    saved = os.dup(1)
    try:
        os.dup2(2, 1)
        yield
    finally:
        sys.stdout.flush()
        os.dup2(saved, 1)
        os.close(saved)


def exclusive_stream(device=None):
    """A torch stream with a hardware queue of its own (ses_stream_create_exclusive): for HipES handles that share a device and
    exchange over the peer-store transport from ONE process -- their kernels wait for each other, so two of them must never sit
    behind one another on one queue.  The stream lives as long as the process (rigs create a handful)."""
    if not torch.cuda.is_available():
        raise SesError("no HIP device visible to torch")
    device = torch.cuda.current_device() if device is None else int(device)
    raw = ctypes.c_void_p()
    check(_lib.load().ses_stream_create_exclusive(device, ctypes.byref(raw)), "ses_stream_create_exclusive")
    return torch.cuda.ExternalStream(raw.value, device=torch.device("cuda", device))


def destroy_stream(stream):
    """Give back a stream of exclusive_stream() (ses_stream_destroy) once every handle built on it is closed.  A rig that
    ends with such streams alive is fine by itself, but under rocprofv3 the HIP runtime's exit-time teardown of live
    CU-masked streams runs after the profiler has finalised and the process ends in SIGSEGV (profiles/README.md, round 6)."""
    stream.synchronize()
    check(_lib.load().ses_stream_destroy(ctypes.c_void_p(stream.cuda_stream)), "ses_stream_destroy")


class HipES:
    """One handle = one (env, network shape, device, stream).  Mirrors the constructor arguments of
    the reference's builder.build_env / build_network (builder.py:10-24)."""

    def __init__(self, env_name="CartPole-v1", num_state=4, num_action=2, discrete_action=True, gru=False,
                 pomdp=False, max_step=500, eval_ep_num=5, device=None, lanes_per_env=0, n_agents=1,
                 physics64=False, stream=None):
        lib = _lib.load()
        if not torch.cuda.is_available():
            raise SesError("no HIP device visible to torch: the simple-es hot path needs an MI355X "
                           "(there is no CPU fallback)")
        if env_name not in ENV_IDS:
            raise SesError(f"env {env_name!r} has no device kernel (available: {sorted(k for k in ENV_IDS if k)})")
        device = torch.cuda.current_device() if device is None else int(device)
        self.device = torch.device("cuda", device)
        self.S, self.A = int(num_state), int(num_action)
        self.discrete, self.gru, self.pomdp = bool(discrete_action), bool(gru), bool(pomdp)
        self.max_step, self.E = int(max_step), int(eval_ep_num)
        self.P = param_count(self.S, self.A, self.gru)
        self.env_id = ENV_IDS[env_name]
        self.n_agents = int(n_agents)
        # width of one initial-state row and its reset distribution
        if self.env_id == ENV_SIMPLE_SPREAD:
            self.init_dim, self.init_range = 4 * self.n_agents, (-1.0, 1.0)
        elif self.env_id == ENV_LUNARLANDER:
            self.init_dim, self.init_range = 16, (0.0, 1.0)      # force, terrain heights, noise key (csrc/ses_lander.h)
        elif self.env_id == ENV_BIPEDALWALKER:
            self.init_dim, self.init_range = 4, (0.0, 1.0)       # force uniform, terrain key (2 words), pad (csrc/ses_walker.h)
        else:
            self.init_dim, self.init_range = 4, (-0.05, 0.05)
        cfg = SesConfig(self.env_id, self.S, self.A, int(self.discrete), int(self.gru), int(self.pomdp),
                        self.max_step, self.E, int(device), int(lanes_per_env), self.n_agents, int(bool(physics64)))
        self._lib = lib
        self._h = ctypes.c_void_p()
        with torch.cuda.device(self.device):
            self.stream = torch.cuda.current_stream(self.device) if stream is None else stream   # the handle's launch stream
            check(lib.ses_create(ctypes.byref(cfg), ctypes.c_void_p(self.stream.cuda_stream), ctypes.byref(self._h)),
                  "ses_create")
        # test hook: SES_TUNING="gru_ep_parallel_max=0,gru_mfma_min_e=1" forces a kernel path for every handle of the
        # process (the library reads no environment itself; tests/test_gpu_gru.py reruns the parity suites this way)
        for item in filter(None, os.environ.get("SES_TUNING", "").split(",")):
            name, _, value = item.partition("=")
            self.set_tuning(name.strip(), int(value))

    def set_tuning(self, name, value):
        """ses_set_tuning: choose among the result-identical rollout kernels (include/ses.h lists the knobs)."""
        check(self._lib.ses_set_tuning(self._h, name.encode(), int(value)), "ses_set_tuning")

    def set_stamp(self, dst):
        """ses_set_stamp: dst = pinned host (or device) int64[1] tensor that receives the GPU real-time counter at the end
        of this handle's next rollouts / perturbation launches; None switches the stamping off."""
        if dst is not None and not (isinstance(dst, torch.Tensor) and dst.dtype == torch.int64 and dst.numel() >= 1 and
                                    (dst.is_pinned() if dst.device.type == "cpu" else dst.device == self.device)):
            raise SesError("set_stamp: expected a pinned host or device int64 tensor")
        self._stamp_keepalive = dst
        check(self._lib.ses_set_stamp(self._h, _ptr(dst)), "ses_set_stamp")

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            self._lib.ses_destroy(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- argument validation ------------------------------------------------------------------
    def _chk(self, t, name, dtype, shape, optional=False):
        if t is None:
            if optional:
                return None
            raise SesError(f"{name}: tensor required")
        if not isinstance(t, torch.Tensor):
            raise SesError(f"{name}: expected a torch tensor, got {type(t).__name__}")
        if t.device != self.device:
            raise SesError(f"{name}: on {t.device}, handle is on {self.device}")
        if t.dtype != dtype:
            raise SesError(f"{name}: dtype {t.dtype}, expected {dtype}")
        if not t.is_contiguous():
            raise SesError(f"{name}: must be contiguous")
        if tuple(t.shape) != tuple(shape):
            raise SesError(f"{name}: shape {tuple(t.shape)}, expected {tuple(shape)}")
        return t

    def _chk_best(self, best):
        """float32[1] that receives max(fitness): a tensor on the handle's device, or PINNED host memory -- the kernel
        then stores the value straight into host-visible memory (no device-to-host copy queued between the kernels)."""
        if best is None:
            return None
        if isinstance(best, torch.Tensor) and best.device.type == "cpu":
            if not (best.is_pinned() and best.dtype == torch.float32 and best.numel() == 1 and best.is_contiguous()):
                raise SesError("best: a host tensor must be pinned float32[1]")
            return best
        return self._chk(best, "best", torch.float32, (1,))

    def _check_parent_idx(self, parent_idx, K):
        """-K <= parent_idx < K, or raise.  The check reads the tensor back (a device sync), so its result is cached --
        on the tensor OBJECT (held here, so its address cannot be recycled for another tensor) and its version."""
        last = getattr(self, "_idx_checked", None)
        if last is not None and last[0] is parent_idx and last[1] == parent_idx._version and last[2] == K:
            return
        lo, hi = int(parent_idx.min()), int(parent_idx.max())
        if hi >= K or lo < -K:
            raise SesError(f"parent_idx outside [-{K}, {K})")
        self._idx_checked = (parent_idx, parent_idx._version, K)

    def empty(self, *shape, dtype=torch.float32):
        return torch.empty(*shape, dtype=dtype, device=self.device)

    def zeros(self, *shape, dtype=torch.float32):
        return torch.zeros(*shape, dtype=dtype, device=self.device)

    def alloc_env_soa(self, n, skew_bytes=4096):
        """SoA env-state arrays (x, xd, th, thd, action i32, ret, status i32) carved out of ONE allocation
        with a 4 KiB skew between consecutive arrays.  Seven separately allocated power-of-two-sized arrays
        start on the same HBM channel/bank phase and their 13 concurrent streams collide; the skew spreads
        them (measured on MI355X at 2^24 envs: 5.4 -> 5.97 TB/s for the env-step kernel)."""
        stride = (n * 4 + skew_bytes + 15) // 16 * 16
        pool = torch.zeros(stride * 7 // 4, dtype=torch.float32, device=self.device)
        views = [pool[k * stride // 4: k * stride // 4 + n] for k in range(7)]
        x, xd, th, thd, action, ret, status = views
        return x, xd, th, thd, action.view(torch.int32), ret, status.view(torch.int32)

    def sync(self):
        check(self._lib.ses_sync(self._h), "ses_sync")

    # -- multi-GPU: RCCL communicator owned by the handle (include/ses.h, ses_comm_*) -------------
    @staticmethod
    def comm_unique_id():
        """128 opaque bytes from ncclGetUniqueId; rank 0 creates them, every rank passes them to comm_init."""
        buf = ctypes.create_string_buffer(_lib.COMM_ID_BYTES)
        with _stdout_to_stderr():
            check(_lib.load().ses_comm_unique_id(buf), "ses_comm_unique_id")
        return buf.raw

    def comm_init(self, rank, world, unique_id):
        if len(unique_id) != _lib.COMM_ID_BYTES:
            raise SesError(f"comm_init: unique id must be {_lib.COMM_ID_BYTES} bytes")
        buf = ctypes.create_string_buffer(bytes(unique_id), _lib.COMM_ID_BYTES)
        with _stdout_to_stderr():          # RCCL prints a version banner on stdout when it initialises
            check(self._lib.ses_comm_init(self._h, int(rank), int(world), buf), "ses_comm_init")
        self._route = None

    def comm_info(self):
        """(rank, world, rccl_version_code); world == 0 means the handle has no communicator."""
        r, w, v = ctypes.c_int32(), ctypes.c_int32(), ctypes.c_int32()
        check(self._lib.ses_comm_info(self._h, ctypes.byref(r), ctypes.byref(w), ctypes.byref(v)), "ses_comm_info")
        return r.value, w.value, v.value

    def comm_destroy(self):
        check(self._lib.ses_comm_destroy(self._h), "ses_comm_destroy")
        self._route = None

    def comm_p2p_export(self, rank, world, max_per_rank):
        """ses_comm_p2p_export: allocate this rank's mailbox of the peer-store transport; returns the 64 handle bytes the
        peers need (gather them over the control plane and pass all of them, in rank order, to comm_p2p_attach)."""
        buf = ctypes.create_string_buffer(_lib.COMM_P2P_HANDLE_BYTES)
        check(self._lib.ses_comm_p2p_export(self._h, int(rank), int(world), int(max_per_rank), buf), "ses_comm_p2p_export")
        return buf.raw

    def comm_p2p_attach(self, handles):
        blob = b"".join(bytes(h) for h in handles)
        check(self._lib.ses_comm_p2p_attach(self._h, ctypes.create_string_buffer(blob, len(blob))), "ses_comm_p2p_attach")
        self._route = None

    def comm_p2p_attach_local(self, peers):
        """ses_comm_p2p_attach_local: `peers` = the HipES handles of all ranks of this process, in rank order (each has
        exported its mailbox and has a stream of its own)."""
        arr = (ctypes.c_void_p * len(peers))(*[p._h.value for p in peers])
        check(self._lib.ses_comm_p2p_attach_local(self._h, arr), "ses_comm_p2p_attach_local")
        self._route = None

    def comm_p2p_info(self):
        """(world, max_per_rank, exchanges); world == 0 means the peer-store transport is not attached."""
        w, m, x = ctypes.c_int32(), ctypes.c_int32(), ctypes.c_int32()
        check(self._lib.ses_comm_p2p_info(self._h, ctypes.byref(w), ctypes.byref(m), ctypes.byref(x)), "ses_comm_p2p_info")
        return w.value, m.value, x.value

    def comm_p2p_counts(self):
        """(exchanges with sequence words, granule exchanges) issued over the attached peer-store transport so far."""
        f, g = ctypes.c_int32(), ctypes.c_int32()
        check(self._lib.ses_comm_p2p_counts(self._h, ctypes.byref(f), ctypes.byref(g)), "ses_comm_p2p_counts")
        return f.value, g.value

    def comm_p2p_status(self):
        """Bit mask of the ranks some peer-store exchange of this handle gave up waiting for (0 = all good).  Reads a
        host-visible word: no stream operation, no synchronisation."""
        m = ctypes.c_uint32()
        check(self._lib.ses_comm_p2p_status(self._h, ctypes.byref(m)), "ses_comm_p2p_status")
        return m.value

    def comm_p2p_reset_status(self):
        """ses_comm_p2p_reset_status: clear the time-out mask (the ranks have agreed to keep using the transport)."""
        check(self._lib.ses_comm_p2p_reset_status(self._h), "ses_comm_p2p_reset_status")

    def comm_p2p_detach(self):
        check(self._lib.ses_comm_p2p_detach(self._h), "ses_comm_p2p_detach")
        self._route = None

    def comm_route(self):
        """(p2p world, p2p floats per rank, rccl world) -- what ses_allgather_fitness has to work with.  Cached: the three
        ctypes calls cost ~2 us per generation otherwise; attach / detach / comm_init / comm_destroy drop the cache."""
        r = getattr(self, "_route", None)
        if r is None:
            w, cap, _ = self.comm_p2p_info()
            r = self._route = (w, cap, self.comm_info()[1])
        return r

    def allgather_fitness(self, local, out=None):
        """local float32[n_per_rank] on every rank -> float32[world * n_per_rank], rank-major, identical everywhere.
        Peer stores when that transport is attached and the shard fits its mailbox, RCCL otherwise."""
        world, cap, rccl_world = self.comm_route()
        if world < 1 or local.shape[0] > cap:
            world = rccl_world                              # (with "comm_force_rccl" both transports span the same ranks)
        if world < 1:
            raise SesError("allgather_fitness: the handle has no communicator (comm_init or comm_p2p_attach first)")
        n = local.shape[0]
        self._chk(local, "local", torch.float32, (n,))
        out = self.empty(world * n) if out is None else self._chk(out, "all", torch.float32, (world * n,))
        check(self._lib.ses_allgather_fitness(self._h, _ptr(local), int(n), _ptr(out)), "ses_allgather_fitness")
        return out

    # -- K1 -----------------------------------------------------------------------------------
    def perturb(self, parents, sigma, seed, gen, first_row, n_rows, parent_idx=None, row_ids=None, out=None,
                idx_in_range=False):
        """idx_in_range: the caller built parent_idx on the host and checked -K <= idx < K there (skips the device
        read-back below)."""
        parents = parents.view(-1, self.P) if parents.dim() == 1 else parents
        K = parents.shape[0]
        self._chk(parents, "parents", torch.float32, (K, self.P))
        self._chk(parent_idx, "parent_idx", torch.int32, (n_rows,), optional=True)
        self._chk(row_ids, "row_ids", torch.int32, (n_rows,), optional=True)
        if parent_idx is not None and not idx_in_range:
            self._check_parent_idx(parent_idx, K)
        theta = self.empty(n_rows, self.P) if out is None else self._chk(out, "theta", torch.float32, (n_rows, self.P))
        check(self._lib.ses_perturb(self._h, _ptr(parents), _ptr(parent_idx), _ptr(row_ids), float(sigma), int(seed),
                                    int(gen), int(first_row), int(n_rows), _ptr(theta)), "ses_perturb")
        return theta

    def noise(self, seed, gen, first_row, n_rows):
        eps = self.empty(n_rows, self.P)
        check(self._lib.ses_noise(self._h, int(seed), int(gen), int(first_row), int(n_rows), _ptr(eps)), "ses_noise")
        return eps

    def perturb_host_noise(self, parents, eps64, sigma, parent_idx=None, want_eps_store=False):
        parents = parents.view(-1, self.P) if parents.dim() == 1 else parents
        K = parents.shape[0]
        n_rows = eps64.shape[0]
        self._chk(parents, "parents", torch.float32, (K, self.P))
        self._chk(eps64, "eps64", torch.float64, (n_rows, self.P))
        self._chk(parent_idx, "parent_idx", torch.int32, (n_rows,), optional=True)
        if parent_idx is not None:
            self._check_parent_idx(parent_idx, K)
        theta = self.empty(n_rows, self.P)
        store = self.empty(n_rows, self.P) if want_eps_store else None
        check(self._lib.ses_perturb_host_noise(self._h, _ptr(parents), _ptr(parent_idx), _ptr(eps64), float(sigma),
                                               int(n_rows), _ptr(theta), _ptr(store)), "ses_perturb_host_noise")
        return (theta, store) if want_eps_store else theta

    def init_states_uniform(self, seed, gen, first_row, n_rows, shared=False, lo=None, hi=None, out=None):
        lo = self.init_range[0] if lo is None else lo
        hi = self.init_range[1] if hi is None else hi
        out = (self.empty(n_rows, self.E, self.init_dim) if out is None else
               self._chk(out, "out", torch.float32, (n_rows, self.E, self.init_dim)))
        check(self._lib.ses_init_states_uniform(self._h, int(seed), int(gen), int(first_row), int(n_rows),
                                                int(bool(shared)), int(self.init_dim), float(lo), float(hi), _ptr(out)),
              "ses_init_states_uniform")
        return out

    def init_states_uniform_gens(self, seed, gen0, gens, first_row, n_rows, shared=False, out=None):
        """The resets of `gens` consecutive generations in one launch: float32[gens, n_rows, E, init_dim]."""
        lo, hi = self.init_range
        out = (self.empty(gens, n_rows, self.E, self.init_dim) if out is None else
               self._chk(out, "out", torch.float32, (gens, n_rows, self.E, self.init_dim)))
        check(self._lib.ses_init_states_uniform_gens(self._h, int(seed), int(gen0), int(gens), int(first_row), int(n_rows),
                                                     int(bool(shared)), int(self.init_dim), float(lo), float(hi), _ptr(out)),
              "ses_init_states_uniform_gens")
        return out

    # -- K2 / K3 ------------------------------------------------------------------------------
    def policy_forward(self, theta, obs, hidden=None):
        n = obs.shape[0]
        self._chk(theta, "theta", torch.float32, (n, self.P))
        self._chk(obs, "obs", torch.float32, (n, self.S))
        if self.gru:
            self._chk(hidden, "hidden", torch.float32, (n, HIDDEN))
        logits = self.empty(n, self.A)
        act = self.empty(n, self.A)
        action = self.empty(n, dtype=torch.int32)
        check(self._lib.ses_policy_forward(self._h, _ptr(theta), _ptr(obs), _ptr(hidden if self.gru else None), int(n),
                                           _ptr(logits), _ptr(act), _ptr(action)), "ses_policy_forward")
        return action, logits, act

    def env_step(self, x, xd, th, thd, action, ret, status, mode=MODE_EPISODIC):
        n = x.shape[0]
        for name, t in (("x", x), ("xd", xd), ("th", th), ("thd", thd), ("ret", ret)):
            self._chk(t, name, torch.float32, (n,))
        self._chk(action, "action", torch.int32, (n,))
        self._chk(status, "status", torch.int32, (n,))   # bit pattern of the uint32 status word
        check(self._lib.ses_env_step(self._h, int(n), int(mode), _ptr(x), _ptr(xd), _ptr(th), _ptr(thd), _ptr(action),
                                     _ptr(ret), _ptr(status)), "ses_env_step")

    # -- k generations per call (ses_run_generations) ---------------------------------------------------
    def run_generations(self, state, k, best, stamps=None):
        """Enqueue k whole generations described by `state` (a _lib.SesGenState the caller keeps alive together with the
        tensors it points to).  best: pinned host (or device) float32[>= k]; stamps: pinned int64[>= k, 2] or None."""
        if not (isinstance(best, torch.Tensor) and best.dtype == torch.float32 and best.numel() >= k and best.is_contiguous()
                and (best.is_pinned() if best.device.type == "cpu" else best.device == self.device)):
            raise SesError("run_generations: best must be a pinned host or device float32 tensor of at least k elements")
        if stamps is not None and not (isinstance(stamps, torch.Tensor) and stamps.dtype == torch.int64 and stamps.numel() >= 2 * k
                                       and stamps.is_contiguous()
                                       and (stamps.is_pinned() if stamps.device.type == "cpu" else stamps.device == self.device)):
            raise SesError("run_generations: stamps must be a pinned host or device int64 tensor of at least 2 k elements")
        check(self._lib.ses_run_generations(self._h, ctypes.byref(state), int(k), _ptr(best), _ptr(stamps)), "ses_run_generations")

    # -- step-wise envs (ses_env_reset / ses_env_step_generic) --------------------------------------
    def env_state_bytes(self):
        b = self._lib.ses_env_state_bytes(self._h)
        if b <= 0:
            check(b, "ses_env_state_bytes")
        return b

    def env_obs_width(self):
        w = self._lib.ses_env_obs_width(self._h)
        if w <= 0:
            check(w, "ses_env_obs_width")
        return w

    def env_reset(self, init):
        """init float32[n, init_dim] -> (state blob uint8[n, state_bytes], obs float32[n, obs_width])."""
        n = init.shape[0]
        self._chk(init, "init", torch.float32, (n, self.init_dim))
        state = torch.zeros(n, self.env_state_bytes(), dtype=torch.uint8, device=self.device)
        obs = self.empty(n, self.env_obs_width())
        check(self._lib.ses_env_reset(self._h, _ptr(init), int(n), _ptr(state), _ptr(obs)), "ses_env_reset")
        return state, obs

    def env_step_generic(self, state, action):
        """One transition of n envs: (obs float32[n, obs_width], reward float32[n], done int32[n]); the state blob is
        updated in place.  action: int32[n] (CartPole), int32[n, n_agents] (simple_spread), float32[n, A] (Box2D envs)."""
        n = state.shape[0]
        self._chk(state, "state", torch.uint8, (n, self.env_state_bytes()))
        if self.env_id == ENV_CARTPOLE:
            self._chk(action, "action", torch.int32, (n,))
        elif self.env_id == ENV_SIMPLE_SPREAD:
            self._chk(action, "action", torch.int32, (n, self.n_agents))
        else:
            self._chk(action, "action", torch.float32, (n, self.A))
        obs = self.empty(n, self.env_obs_width())
        reward = self.empty(n)
        done = self.empty(n, dtype=torch.int32)
        check(self._lib.ses_env_step_generic(self._h, _ptr(state), _ptr(action), int(n), _ptr(obs), _ptr(reward), _ptr(done)),
              "ses_env_step_generic")
        return obs, reward, done

    def env_step_shape(self):
        """ses_env_step_shape: (threads per workgroup, LDS bytes reserved per workgroup, waves per CU by the occupancy
        calculator, the device's LDS bytes per CU)."""
        b, l, w, c = ctypes.c_int32(), ctypes.c_int32(), ctypes.c_int32(), ctypes.c_int32()
        check(self._lib.ses_env_step_shape(self._h, ctypes.byref(b), ctypes.byref(l), ctypes.byref(w), ctypes.byref(c)),
              "ses_env_step_shape")
        return b.value, l.value, w.value, c.value

    def stream_probe(self, x, xd, th, thd, action, ret, status):
        """ses_stream_probe: the env-step kernel's 13 streams with no arithmetic (values unchanged) -- bench.py's ceiling."""
        n = x.shape[0]
        for name, t in (("x", x), ("xd", xd), ("th", th), ("thd", thd), ("ret", ret)):
            self._chk(t, name, torch.float32, (n,))
        self._chk(action, "action", torch.int32, (n,))
        self._chk(status, "status", torch.int32, (n,))
        check(self._lib.ses_stream_probe(self._h, int(n), _ptr(x), _ptr(xd), _ptr(th), _ptr(thd), _ptr(action), _ptr(ret),
                                         _ptr(status)), "ses_stream_probe")

    # -- fused rollout ------------------------------------------------------------------------
    def rollout(self, theta, init, mode=MODE_EPISODIC, want_episodes=False, fitness=None):
        n_rows = theta.shape[0]
        self._chk(theta, "theta", torch.float32, (n_rows, self.P))
        if init.dim() == 2:
            self._chk(init, "init", torch.float32, (self.E, self.init_dim))
            per = 0
        else:
            self._chk(init, "init", torch.float32, (n_rows, self.E, self.init_dim))
            per = 1
        fitness = self.empty(n_rows) if fitness is None else self._chk(fitness, "fitness", torch.float32, (n_rows,))
        ep_ret = self.empty(n_rows, self.E, dtype=torch.float64) if want_episodes else None
        ep_steps = self.empty(n_rows, self.E, dtype=torch.int32) if (want_episodes and self.env_id != ENV_SIMPLE_SPREAD) else None
        check(self._lib.ses_rollout(self._h, _ptr(theta), _ptr(init), per, int(n_rows), int(mode), _ptr(fitness),
                                    _ptr(ep_ret), _ptr(ep_steps)), "ses_rollout")
        return (fitness, ep_ret, ep_steps) if want_episodes else fitness

    # -- K4 / K5 / K6 -------------------------------------------------------------------------
    def rank_center(self, fitness, want_weights=True, best=None):
        """rank int32[n], weights float64[n] (None when want_weights is False).  best: optional float32[1] device
        tensor that receives max(fitness) in the same launch."""
        n = fitness.shape[0]
        self._chk(fitness, "fitness", torch.float32, (n,))
        self._chk_best(best)
        rank = self.empty(n, dtype=torch.int32)
        weights = self.empty(n, dtype=torch.float64) if want_weights else None
        check(self._lib.ses_rank_center(self._h, _ptr(fitness), int(n), _ptr(rank), _ptr(weights), _ptr(best)),
              "ses_rank_center")
        return rank, weights

    def es_update_philox(self, weights, seed, gen, lr, sigma, adam_a, mu, m, v, skip_row0=True, want_grad=False):
        n = weights.shape[0]
        self._chk(weights, "weights", torch.float64, (n,))
        for name, t in (("mu", mu), ("m", m), ("v", v)):
            self._chk(t, name, torch.float32, (self.P,))
        grad = self.empty(self.P) if want_grad else None
        check(self._lib.ses_es_update_philox(self._h, _ptr(weights), int(n), int(bool(skip_row0)), int(seed), int(gen),
                                             float(lr), float(sigma), float(adam_a), _ptr(mu), _ptr(m), _ptr(v),
                                             _ptr(grad)), "ses_es_update_philox")
        return grad

    def openai_sharded_ok(self, comm, n, per_rank, world):
        """ses_openai_sharded_ok: can the openai_es tail of this layout run in its shard form over `comm`'s transport?"""
        rc = self._lib.ses_openai_sharded_ok(self._h, comm._h, int(n), int(per_rank), int(world))
        if rc < 0:
            check(rc, "ses_openai_sharded_ok")
        return rc == 1

    def openai_generation(self, fitness, seed, gen, lr, sigma, adam_a, state_in, state_out, next_sigma, next_gen,
                          first_row, n_rows, theta_next=None, best=None, comm=None, per_rank=0, world=1):
        """ses_openai_generation: rank shaping + ES gradient + Adam + the next population in four launches (five above 8192 rows).
        state_in / state_out: (mu, m, v) triples of distinct float32[P] tensors.  Returns theta_next[n_rows, P].
        comm (a HipES that owns a transport of `world` ranks) selects ses_openai_generation_sharded: this rank ranks and
        accumulates its own rows only, the chunk partials are all-gathered over comm (openai_sharded_ok says when)."""
        n = fitness.shape[0]
        self._chk(fitness, "fitness", torch.float32, (n,))
        for name, t in zip(("mu_in", "m_in", "v_in", "mu_out", "m_out", "v_out"), tuple(state_in) + tuple(state_out)):
            self._chk(t, name, torch.float32, (self.P,))
        if any(a.data_ptr() == b.data_ptr() for a, b in zip(state_in, state_out)):
            raise SesError("openai_generation: state_in and state_out must be distinct buffers")
        self._chk_best(best)
        if not (0 <= first_row and first_row + n_rows <= n):
            raise SesError(f"openai_generation: rows [{first_row}, +{n_rows}) outside the population of {n}")
        theta = (self.empty(n_rows, self.P) if theta_next is None else
                 self._chk(theta_next, "theta_next", torch.float32, (n_rows, self.P)))
        if comm is not None:
            check(self._lib.ses_openai_generation_sharded(
                self._h, comm._h, _ptr(fitness), int(n), int(seed), int(gen), float(lr), float(sigma), float(adam_a),
                *[_ptr(t) for t in state_in], *[_ptr(t) for t in state_out], float(next_sigma), int(next_gen), int(first_row),
                int(n_rows), int(per_rank), int(world), _ptr(theta), _ptr(best)), "ses_openai_generation_sharded")
            return theta
        check(self._lib.ses_openai_generation(self._h, _ptr(fitness), int(n), int(seed), int(gen), float(lr), float(sigma),
                                              float(adam_a), *[_ptr(t) for t in state_in], *[_ptr(t) for t in state_out],
                                              float(next_sigma), int(next_gen), int(first_row), int(n_rows),
                                              _ptr(theta) if n_rows else None, _ptr(best)), "ses_openai_generation")
        return theta

    def es_update_stored(self, weights, eps_store, lr, sigma, adam_a, mu, m, v, want_grad=False):
        n = weights.shape[0]
        self._chk(weights, "weights", torch.float64, (n,))
        self._chk(eps_store, "eps_store", torch.float32, (n, self.P))
        for name, t in (("mu", mu), ("m", m), ("v", v)):
            self._chk(t, name, torch.float32, (self.P,))
        grad = self.empty(self.P) if want_grad else None
        check(self._lib.ses_es_update_stored(self._h, _ptr(weights), int(n), _ptr(eps_store), float(lr), float(sigma),
                                             float(adam_a), _ptr(mu), _ptr(m), _ptr(v), _ptr(grad)),
              "ses_es_update_stored")
        return grad

    def elite_ids(self, rank, k):
        n = rank.shape[0]
        self._chk(rank, "rank", torch.int32, (n,))
        if not 1 <= k <= n:
            raise SesError(f"elite_ids: need 1 <= k <= n, got k={k} n={n}")
        ids = self.empty(k, dtype=torch.int32)
        check(self._lib.ses_elite_ids(self._h, _ptr(rank), int(n), int(k), _ptr(ids)), "ses_elite_ids")
        return ids

    def elite_select(self, rank, k, parent_map, alias_state=None):
        """One launch: (elite_ids[k], elite_parent_idx[k], alias_first[k] or None), all on the device.
        alias_state: int32[1] device tensor updated in place (simple_evolution), or None (simple_genetic)."""
        n = rank.shape[0]
        self._chk(rank, "rank", torch.int32, (n,))
        self._chk(parent_map, "parent_map", torch.int32, (n,))
        self._chk(alias_state, "alias_state", torch.int32, (1,), optional=True)
        if not 1 <= k <= min(n, 1024):
            raise SesError(f"elite_select: need 1 <= k <= min(n, 1024), got k={k} n={n}")
        out = self.empty(3, k, dtype=torch.int32)
        alias = out[2] if alias_state is not None else None
        check(self._lib.ses_elite_select(self._h, _ptr(rank), int(n), int(k), _ptr(parent_map), _ptr(alias_state),
                                         _ptr(out[0]), _ptr(out[1]), _ptr(alias)), "ses_elite_select")
        return out[0], out[1], alias

    def elite_mean(self, rows, alias_first=None):
        k = rows.shape[0]
        self._chk(rows, "rows", torch.float32, (k, self.P))
        self._chk(alias_first, "alias_first", torch.int32, (k,), optional=True)
        mean = self.empty(self.P)
        check(self._lib.ses_elite_mean(self._h, _ptr(rows), _ptr(alias_first), int(k), _ptr(mean)), "ses_elite_mean")
        return mean

    def gather_rows(self, src, ids):
        n_src = src.shape[0]
        k = ids.shape[0]
        self._chk(src, "src", torch.float32, (n_src, self.P))
        self._chk(ids, "ids", torch.int32, (k,))
        lo, hi = int(ids.min()), int(ids.max())
        if lo < 0 or hi >= n_src:
            raise SesError(f"gather_rows: ids outside [0, {n_src})")
        dst = self.empty(k, self.P)
        check(self._lib.ses_gather_rows(self._h, _ptr(src), _ptr(ids), int(k), _ptr(dst)), "ses_gather_rows")
        return dst
