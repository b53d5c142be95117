"""Population sharding over the GPUs of one node (one process per GPU, torch.distributed; backend "nccl"
is RCCL on ROCm, "gloo" in the CPU tests).

Replaces the reference's only parallelism, `multiprocessing.Pool.map` over offspring
(learning_strategies/evolution/loop.py:66-79).  Offspring are independent, so rank r owns the contiguous
rows [first, first + n_local) of the global population; the single exchange step per generation is an
all-gather of the per-offspring fitness (N * 4 bytes: latency-bound over xGMI, no bucketing needed).
Noise is counter-based on the GLOBAL row index, so every rank can regenerate any row and computes the
identical parent update without a second collective.
"""
import torch
import torch.distributed as dist


class Shard:
    def __init__(self, n_global, group=None):
        self.group = group
        on = dist.is_available() and dist.is_initialized()
        self.world = dist.get_world_size(group) if on else 1
        self.rank = dist.get_rank(group) if on else 0
        self.n_global = int(n_global)
        self.per_rank = -(-self.n_global // self.world)                 # ceil: equal-sized all-gather slots
        self.first = min(self.rank * self.per_rank, self.n_global)
        self.n_local = max(0, min(self.per_rank, self.n_global - self.first))

    def allgather_fitness(self, local):
        """local: float32[n_local] on this rank's device -> float32[n_global], identical on every rank."""
        if self.world == 1:
            return local
        slot = local
        if self.n_local != self.per_rank:                                # ragged tail: pad the last rank(s)
            slot = local.new_full((self.per_rank,), float("-inf"))
            slot[: self.n_local] = local
        out = local.new_empty(self.per_rank * self.world)
        if local.is_cuda and dist.get_backend(self.group) == "gloo":
            # test rigs only (several ranks sharing one GPU, where RCCL refuses duplicate devices): stage the
            # 4*N bytes through the host.  Production is backend "nccl" (RCCL) on device buffers.
            host = torch.empty(out.shape, dtype=out.dtype)
            dist.all_gather_into_tensor(host, slot.contiguous().cpu(), group=self.group)
            out.copy_(host)
        else:
            dist.all_gather_into_tensor(out, slot.contiguous(), group=self.group)
        return out[: self.n_global].contiguous()

    def barrier(self):
        if self.world > 1:
            dist.barrier(group=self.group)
