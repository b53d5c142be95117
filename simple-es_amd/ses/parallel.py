"""Population sharding over the GPUs of one node, one process per GPU.

Replaces the reference's only parallelism, `multiprocessing.Pool.map` over offspring
(learning_strategies/evolution/loop.py:66-79).  Offspring are independent, so rank r owns the contiguous
rows [first, first + n_local) of the global population; the single exchange step per generation is an
all-gather of the per-offspring fitness (N * 4 bytes: latency-bound over xGMI, no bucketing needed).
Noise is counter-based on the GLOBAL row index, so every rank can regenerate any row and computes the
identical parent update without a second collective.

The collective itself lives in the C library (`ses_allgather_fitness` on the handle's stream: peer stores into mapped
mailboxes inside a node, RCCL otherwise): `attach_comm(dev)` sets the transports up, using torch.distributed only as the
control plane that carries the mailbox handles / rank 0's 128-byte RCCL id between the ranks.  The torch.distributed
route further down is what is left when neither is available (CPU tensors, or SES_COMM_P2P=0 with backend "gloo").
"""
import contextlib
import os
import socket
import sys

import torch
import torch.distributed as dist


_SOLO = 0      # > 0 inside `with solo():` -- this process then behaves like a single-rank run


def _dist_on():
    return _SOLO == 0 and dist.is_available() and dist.is_initialized()


def world_size(group=None):
    """Ranks the population is sharded over: the process group's size, 1 without one or inside `with solo():`."""
    return dist.get_world_size(group) if _dist_on() else 1


@contextlib.contextmanager
def solo():
    """Inside the block this process is a world of ONE, whatever process group exists: Shard() owns every row, attach_comm()
    is a no-op and ESLoop / the strategies take their single-GPU paths.  For a rank that has to RECOMPUTE on its own what the
    sharded run produced -- bench.py's `shard_check`, the multi-GPU tests -- without leaving the process group.  Objects built
    inside the block stay single-rank afterwards (their Shard is fixed at construction); the peers must not be inside a
    collective that waits for this rank meanwhile."""
    global _SOLO
    _SOLO += 1
    try:
        yield
    finally:
        _SOLO -= 1


_COMM = {}   # (group, world) -> HipES handle that owns this process's communicator(s), or False (torch route)
P2P_MAX_PER_RANK = 65536          # floats per rank a mailbox slot holds (256 KB): every population this repo shards fits


def _cpu_group_ok(ok, device, group):
    """MIN over the ranks of a 0/1 flag (one rank failed -> nobody uses the transport)."""
    backend = dist.get_backend(group)
    flag = torch.tensor([int(ok)], dtype=torch.int32, device=device if backend == "nccl" else "cpu")
    dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=group)
    return int(flag.item()) == 1


def all_ranks(ok, device, group=None):
    """COLLECTIVE: True iff `ok` is true on every rank of the group (world 1 / no process group: `ok` itself)."""
    if not _dist_on() or dist.get_world_size(group) == 1:
        return bool(ok)
    return _cpu_group_ok(ok, device, group)


def _attach_p2p(owner, rank, world, group):
    """Peer-store transport (ses_comm_p2p_*): every rank exports a mailbox, the handles travel over the control plane,
    every rank maps the others; then ONE exchange of a known pattern is checked on every rank.  All ranks must be on one
    node.  Any failure on any rank -> every rank detaches (collective decision)."""
    if os.environ.get("SES_COMM_P2P", "1") == "0" or not 2 <= world <= 16:
        return False
    hosts = [None] * world
    dist.all_gather_object(hosts, socket.gethostname(), group=group)
    ok, handle = len(set(hosts)) == 1, None
    if ok:
        try:
            handle = owner.comm_p2p_export(rank, world, P2P_MAX_PER_RANK)
        except Exception as exc:
            ok = False
            print(f"[ses] rank {rank}: peer-store mailbox not available ({exc})", file=sys.stderr, flush=True)
    handles = [None] * world
    dist.all_gather_object(handles, handle, group=group)
    ok = ok and all(h is not None for h in handles)
    if ok:
        try:
            owner.comm_p2p_attach(handles)
        except Exception as exc:
            ok = False
            print(f"[ses] rank {rank}: peer mailboxes not mappable ({exc})", file=sys.stderr, flush=True)
    ok = _cpu_group_ok(ok, owner.device, group)            # also: nobody starts the test before everybody has attached
    if ok:
        try:
            # ("comm_p2p_keep_going" stays 0 here: the exchange AFTER a timed-out one fails loudly with SES_ERR_COMM.  Only a
            # caller that owns a recovery -- ESLoop.run() for a guarded run, comm_keep_going below -- switches it on.)
            # The two checked exchanges below wait 5 s at most for a peer (not the run's time-out, 60 s by default): every rank is
            # HERE, a peer that does not answer within seconds will not answer at all.
            owner.set_tuning("comm_p2p_timeout_ms", 5000)
            n = 257
            mine = torch.arange(n, device=owner.device, dtype=torch.float32) + 1000.0 * (rank + 1)
            got = owner.allgather_fitness(mine).cpu()
            owner.sync()
            want = torch.cat([torch.arange(n, dtype=torch.float32) + 1000.0 * (r + 1) for r in range(world)])
            ok = bool(torch.equal(got, want))
        except Exception as exc:
            ok = False
            print(f"[ses] rank {rank}: peer-store self-test failed ({exc})", file=sys.stderr, flush=True)
        ok = _cpu_group_ok(ok, owner.device, group)
    if ok:
        # The granule exchanges (8-byte {exchange number, value} stores that are their own flag: the chunk partials of the shard
        # form of the openai_es tail) are checked separately, with a short time-out: if they fail on ANY rank they are switched
        # off on every rank -- the flag-based exchanges, which have just been checked, then carry the partials too.
        good = True
        try:
            owner.set_tuning("comm_granule_allgather", 1)
            n = 1031
            mine = torch.arange(n, device=owner.device, dtype=torch.float32) * 0.5 - 7.0 * (rank + 1)
            got = owner.allgather_fitness(mine).cpu()
            owner.sync()
            want = torch.cat([torch.arange(n, dtype=torch.float32) * 0.5 - 7.0 * (r + 1) for r in range(world)])
            good = bool(torch.equal(got, want)) and owner.comm_p2p_status() == 0
        except Exception as exc:
            good = False
            print(f"[ses] rank {rank}: granule exchange self-test failed ({exc})", file=sys.stderr, flush=True)
        finally:
            owner.set_tuning("comm_granule_allgather", 0)
        good = _cpu_group_ok(good, owner.device, group)
        owner.set_tuning("comm_p2p_timeout_ms", int(os.environ.get("SES_COMM_P2P_TIMEOUT_MS", "0")))    # the run's own (0 = 60 s)
        if not good:
            print(f"[ses] rank {rank}: granule exchanges switched off on this transport (flag-based exchanges carry everything)",
                  file=sys.stderr, flush=True)
            owner.set_tuning("comm_granules_enabled", 0)
            owner.comm_p2p_reset_status()
            dist.barrier(group=group)
    if not ok:
        try:
            owner.comm_p2p_detach()
        except Exception:
            pass
    return ok


def _attach_rccl(owner, rank, world, group):
    from .device import HipES
    ok = 1
    try:
        box = [HipES.comm_unique_id() if rank == 0 else None]
    except Exception as exc:                         # librccl not loadable, ...
        ok, box = 0, [None]
        print(f"[ses] rank {rank}: RCCL unavailable ({exc}); falling back to torch.distributed", flush=True)
    dist.broadcast_object_list(box, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
    if ok and box[0] is not None:
        try:
            owner.comm_init(rank, world, box[0])
        except Exception as exc:
            ok = 0
            print(f"[ses] rank {rank}: ses_comm_init failed ({exc}); falling back to torch.distributed", flush=True)
    else:
        ok = 0
    agreed = _cpu_group_ok(ok, owner.device, group)
    if not agreed and ok:
        owner.comm_destroy()
    return agreed


def attach_comm(dev, group=None, allow_single=False, create=True):
    """Give the HipES handle `dev` access to the library's all-gather over the process group (no-op at world 1).
    Collective: every rank of the group must call it.  One dedicated long-lived handle per process and group, bound to
    the same stream, owns the transports and is shared by every loop of the process (bench.py builds several):
      * the peer-store transport (ranks on one node, any backend -- ranks may even share a GPU, which is how the
        single-GPU tests drive it); verified by one checked exchange before it is used;
      * an RCCL communicator when the backend is "nccl" (shards beyond the mailbox size, or no peer stores).
    If neither can be set up on ALL ranks, every rank falls back to torch.distributed's all-gather, which is reported
    (comm_transport() == "torch") instead of failing the run.  Returns True when the library carries the data path.
    create=False: NOT collective -- bind `dev` to the owner this process already has for the group, or to none (a strategy
    object built on a subset of the ranks, e.g. a rank-0 evaluation script under torchrun, must not start a rendezvous)."""
    if not _dist_on() or (dist.get_world_size(group) == 1 and not allow_single):   # allow_single: the 1-GPU test of this path
        return False
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    key = (id(group) if group is not None else 0, world)
    owner = _COMM.get(key)
    if owner is None and not create:
        dev._comm_owner = None
        return False
    if owner is None:
        from .device import HipES
        owner = HipES(None, dev.S, dev.A, dev.discrete, dev.gru, device=dev.device.index)
        have_p2p = _attach_p2p(owner, rank, world, group)
        have_rccl = dist.get_backend(group) == "nccl" and _attach_rccl(owner, rank, world, group)   # ranks sharing a GPU: never
        if not (have_p2p or have_rccl):
            owner.close()
            owner = False
        _COMM[key] = owner
    if owner is False or owner.stream.cuda_stream != dev.stream.cuda_stream:
        dev._comm_owner = None
        return False
    dev._comm_owner = owner
    return True


def comm_keep_going(dev, on):
    """Let `dev`'s peer-store exchanges continue after a time-out (NaN-filled shards, status word set) instead of failing the
    next call.  ONLY for a caller that polls comm_failed / comm_recover and rolls back: ESLoop.run() switches it on for a
    guarded run and off again when the run ends; everybody else (RolloutWorker, bench legs, strategies with
    noise='numpy', user code calling attach_comm) keeps the loud failure."""
    owner = getattr(dev, "_comm_owner", None)
    if owner is not None and owner.comm_route()[0]:
        owner.set_tuning("comm_p2p_keep_going", 1 if on else 0)


def comm_failed(dev):
    """True when a peer-store exchange of this process has given up waiting for a peer (local view, no stream operation)."""
    owner = getattr(dev, "_comm_owner", None)
    return bool(owner is not None and owner.comm_route()[0] and owner.comm_p2p_status())


def comm_recover(dev, group=None):
    """COLLECTIVE check of the peer-store transport: did any rank's exchange time out since the last check?  If so every
    rank drains its stream, drops the transport (ses_comm_p2p_detach) and the all-gather continues on RCCL or, without a
    communicator, on torch.distributed.  Returns True when that happened: the fitness vectors since the last check may
    hold NaN shards on some ranks, so the caller must roll back to the state it had then (ESLoop.run does, and replays
    the generations over the fallback transport: the noise is counter-based, so the replay reproduces them bit for bit)."""
    owner = getattr(dev, "_comm_owner", None)
    if not _dist_on() or dist.get_world_size(group) == 1 or owner is None or not owner.comm_route()[0]:
        return False
    all_good = _cpu_group_ok(not owner.comm_p2p_status(), owner.device, group)
    if all_good:
        return False
    torch.cuda.synchronize(owner.device)           # exchanges in flight end by themselves (time-out at the latest)
    mask = owner.comm_p2p_status()
    dist.barrier(group=group)                      # every rank's kernels are done: nobody frees a mailbox a peer still stores into
    owner.comm_p2p_detach()
    dist.barrier(group=group)
    nxt = "RCCL" if owner.comm_route()[2] else "torch.distributed"
    print(f"[ses] rank {dist.get_rank(group)}: a peer-store exchange timed out (local mask 0x{mask:x}); the fitness "
          f"all-gather continues on {nxt}", file=sys.stderr, flush=True)
    return True


def comm_info(dev):
    """(rank, world, rccl_version) of the RCCL communicator behind `dev`'s all-gathers; world 0 = none."""
    owner = getattr(dev, "_comm_owner", None) or dev
    return owner.comm_info()


def comm_transport(dev, per_rank=1):
    """What carries `dev`'s fitness all-gather for shards of per_rank floats: "p2p-store", "rccl" or "torch"."""
    owner = getattr(dev, "_comm_owner", None)
    if owner is None:
        return "torch"
    world, cap, rccl_world = owner.comm_route()
    if world and per_rank <= cap:
        return "p2p-store"
    return "rccl" if rccl_world else "torch"


class Shard:
    def __init__(self, n_global, group=None):
        self.group = group
        on = _dist_on()
        self.world = dist.get_world_size(group) if on else 1
        self.rank = dist.get_rank(group) if on else 0
        self.n_global = int(n_global)
        self.per_rank = -(-self.n_global // self.world)                 # ceil: equal-sized all-gather slots
        self.first = min(self.rank * self.per_rank, self.n_global)
        self.n_local = max(0, min(self.per_rank, self.n_global - self.first))

    def allgather_fitness(self, local, dev=None):
        """local: float32[n_local] on this rank's device -> float32[n_global], identical on every rank.
        dev: the HipES handle whose communicator (attach_comm) carries the collective."""
        if self.world == 1:
            return local
        slot = local
        if self.n_local != self.per_rank:                                # ragged tail: pad the last rank(s)
            slot = local.new_full((self.per_rank,), float("-inf"))
            slot[: self.n_local] = local
        owner = getattr(dev, "_comm_owner", None) if dev is not None else None
        if owner is None and dev is not None and dev.comm_route()[2] == self.world:
            owner = dev                                                  # a handle that was given its own communicator
        route = owner.comm_route() if owner is not None else (0, 0, 0)   # cached on the handle: no ctypes call per generation
        if (route[0] == self.world and self.per_rank <= route[1]) or route[2] == self.world:
            out = owner.allgather_fitness(slot.contiguous())            # ses_allgather_fitness: peer stores or RCCL
        elif local.is_cuda and dist.get_backend(self.group) == "gloo":
            # test rigs only (several ranks sharing one GPU): stage the 4*N bytes through the host
            out = local.new_empty(self.per_rank * self.world)
            host = torch.empty(out.shape, dtype=out.dtype)
            dist.all_gather_into_tensor(host, slot.contiguous().cpu(), group=self.group)
            out.copy_(host)
        else:
            out = local.new_empty(self.per_rank * self.world)
            dist.all_gather_into_tensor(out, slot.contiguous(), group=self.group)
        return out if out.shape[0] == self.n_global else out[: self.n_global].contiguous()

    def barrier(self):
        if self.world > 1:
            dist.barrier(group=self.group)
