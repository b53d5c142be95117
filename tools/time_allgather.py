#!/usr/bin/env python3
"""Latency of ses_allgather_fitness over the peer-store transport, ranks sharing ONE GPU (development rig: several
processes on one device, mailboxes mapped with hipIpc exactly as across GPUs, but no xGMI hop).
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 4 --master-addr 127.0.0.1 --master-port P tools/time_allgather.py
Each rank enqueues `reps` back-to-back exchanges of 4096 floats and reports the HIP-event time per exchange (rank 0 prints)."""
import json, os, sys, statistics
import torch, torch.distributed as dist
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "simple-es_amd"))
from ses import HipES
from ses.parallel import attach_comm, comm_transport

dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
es = HipES("CartPole-v1", 4, 2, True, False, max_step=500, eval_ep_num=5)
assert attach_comm(es), "no library transport"
owner = es._comm_owner
for n, gran in ((4096, 1), (4096, 0), (8192, 1), (8192, 0), (65536 // world, 1), (65536 // world, 0)):
    owner.set_tuning("comm_granule_allgather", gran)        # 1: {exchange number, value} granules; 0: data + sequence words
    local = torch.full((n,), float(rank), device=es.device)
    out = es.empty(world * n)
    for _ in range(20):
        owner.allgather_fitness(local, out=out)
    torch.cuda.synchronize(); dist.barrier()
    ts = []
    for rep in range(9):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        with torch.cuda.stream(es.stream):
            e0.record()
            for _ in range(100):
                owner.allgather_fitness(local, out=out)
            e1.record()
        e1.synchronize()
        ts.append(e0.elapsed_time(e1) * 10.0)
        dist.barrier()
    ok = bool(torch.equal(out.view(world, n)[:, 0].cpu(), torch.arange(world, dtype=torch.float32)))
    if rank == 0:
        print(json.dumps({"ranks_on_one_gpu": world, "floats_per_rank": n, "transport": comm_transport(es, n), "granules": bool(gran and n <= 32768),
                          "us_per_exchange": round(statistics.median(ts), 2), "correct": ok}), flush=True)
es.close()
dist.destroy_process_group()
