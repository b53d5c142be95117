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


def attach_comm(dev, group=None):
    """Give the HipES handle `dev` an RCCL communicator spanning the process group (no-op at world 1 or when
    it has one).  Collective: every rank of the group must call it.  Returns True when the handle's
    communicator is the data path, False when torch.distributed's is (gloo test rigs)."""
    if not _dist_on() or dist.get_world_size(group) == 1:
        return False
    if dist.get_backend(group) != "nccl":           # ranks may share a GPU: keep the staged gloo path
        return False
    if dev.comm_info()[1] > 0:
        return True
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    box = [dev.comm_unique_id() if rank == 0 else None]
    dist.broadcast_object_list(box, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
    dev.comm_init(rank, world, box[0])
    return True


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
        if dev is not None and dev.comm_info()[1] == self.world:
            out = dev.allgather_fitness(slot.contiguous())              # ses_allgather_fitness: RCCL, handle's stream
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
