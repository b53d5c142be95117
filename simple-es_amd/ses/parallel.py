"""Population sharding over the GPUs of one node, one process per GPU.

Replaces the reference's only parallelism, `multiprocessing.Pool.map` over offspring
(learning_strategies/evolution/loop.py:66-79).  Offspring are independent, so rank r owns the contiguous
rows [first, first + n_local) of the global population; the single exchange step per generation is an
all-gather of the per-offspring fitness (N * 4 bytes: latency-bound over xGMI, no bucketing needed).
Noise is counter-based on the GLOBAL row index, so every rank can regenerate any row and computes the
identical parent update without a second collective.

The collective itself lives in the C library (`ses_allgather_fitness`, RCCL on the handle's stream):
`attach_comm(dev)` creates the handle's communicator, using torch.distributed only as the control plane that
carries rank 0's 128-byte unique id to the other ranks.  The torch.distributed route further down is the test
rig: backend "gloo" with several ranks sharing one GPU (RCCL refuses duplicate devices) or CPU tensors.
"""
import torch
import torch.distributed as dist


def _dist_on():
    return dist.is_available() and dist.is_initialized()


_COMM = {}   # (group, world) -> HipES handle that owns this process's RCCL communicator, or False (torch route)


def attach_comm(dev, group=None, allow_single=False):
    """Give the HipES handle `dev` access to an RCCL communicator spanning the process group (no-op at world 1).
    Collective: every rank of the group must call it.  One communicator per process and group is created -- on a
    dedicated long-lived handle bound to the same stream -- and shared by every loop of the process (bench.py builds
    several).  If RCCL cannot be initialised on ANY rank, all ranks agree to fall back to torch.distributed's
    all-gather, which is reported (comm_info() world = 0) instead of failing the run.
    Returns True when the library's communicator is the data path."""
    if not _dist_on() or (dist.get_world_size(group) == 1 and not allow_single):   # allow_single: the 1-GPU test of this path
        return False
    if dist.get_backend(group) != "nccl":           # ranks may share a GPU: keep the staged gloo path
        return False
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    key = (id(group) if group is not None else 0, world)
    owner = _COMM.get(key)
    if owner is None:
        from .device import HipES
        ok = 1
        try:
            owner = HipES(None, dev.S, dev.A, dev.discrete, dev.gru, device=dev.device.index)
            box = [HipES.comm_unique_id() if rank == 0 else None]
        except Exception as exc:                     # librccl not loadable, ...
            ok, box, owner = 0, [None], None
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
        flag = torch.tensor([ok], device=dev.device, dtype=torch.int32)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=group)      # one rank failed -> nobody uses the communicator
        if int(flag.item()) == 1:
            _COMM[key] = owner
        else:
            if owner is not None:
                owner.close()
            owner = _COMM[key] = False
    if owner is False or owner.stream.cuda_stream != dev.stream.cuda_stream:
        dev._comm_owner = None
        return False
    dev._comm_owner = owner
    return True


def comm_info(dev):
    """(rank, world, rccl_version) of the communicator that carries `dev`'s all-gathers; world 0 = torch.distributed route."""
    owner = getattr(dev, "_comm_owner", None) or dev
    return owner.comm_info()


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
        if owner is None and dev is not None and dev.comm_info()[1] == self.world:
            owner = dev                                                  # a handle that was given its own communicator
        if owner is not None and owner.comm_info()[1] == self.world:
            out = owner.allgather_fitness(slot.contiguous())            # ses_allgather_fitness: RCCL, handle's stream
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
