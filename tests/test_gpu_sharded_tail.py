"""The shard form of the openai_es tail (ses_openai_generation_sharded) at the world sizes the 1-GPU rig cannot form with
processes: 8 and 16 ranks as HANDLES of one process, each on a stream with a hardware queue of its own, their mailboxes attached directly
(ses_comm_p2p_attach_local) -- the peer-store protocol, the kernels and the layouts are those of one process per GPU.
Every rank ranks and accumulates its own rows only; parent, Adam moments, best reward and the next population's rows must
equal the replicated ses_openai_generation bit for bit (offspring_strategies.py:380-419 + :284-328 of the reference,
evaluated once)."""
import os
import subprocess
import sys
import textwrap

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "simple-es_amd")

WORKER = textwrap.dedent("""
    import os, sys
    world, per, missing, P_gru, granules = (int(v) for v in sys.argv[1:6])
    import numpy as np, torch
    sys.path[:0] = [%r, %r]
    from ses import HipES, exclusive_stream
    n = world * per - missing
    # a hardware queue per rank, by construction (ses_stream_create_exclusive: streams that never enter the runtime's queue
    # pool): an exchange kernel waits for kernels of its peers' streams, so none of them may queue up behind another
    streams = [exclusive_stream() for _ in range(world)]
    ranks = [HipES(None, 4, 2, True, bool(P_gru), stream=streams[r]) for r in range(world)]
    ref = HipES(None, 4, 2, True, bool(P_gru))
    P = ref.P
    for r, es in enumerate(ranks):
        es.set_tuning("comm_p2p_timeout_ms", 5000)
        es.set_tuning("openai_sharded_min_rows", 0)          # (the product keeps populations below 8192 rows replicated: round 6)
        es.set_tuning("openai_granule_exchange", granules)   # 1: the gradient kernel stores {sequence, value} granules into the
                                                             # peers' mailboxes and the update polls them; 0: float all-gather launch
        es.comm_p2p_export(r, world, 65536)
    for es in ranks:
        es.comm_p2p_attach_local(ranks)
        assert es.openai_sharded_ok(es, n, per, world)
    assert not ranks[0].openai_sharded_ok(ranks[0], n, per + 1, world)         # not chunk-aligned
    g = torch.Generator(device="cuda").manual_seed(1)
    state = [torch.randn(P, device="cuda", generator=g) * 0.1, torch.randn(P, device="cuda", generator=g) * 0.01,
             torch.rand(P, device="cuda", generator=g) * 0.01]
    theta_ref = ref.empty(n, P)
    best_ref = ref.empty(1)
    outs = [[es.empty(P) for _ in range(3)] for es in ranks]
    thetas = [es.empty(min(per, n - r * per), P) for r, es in enumerate(ranks)]
    bests = [es.empty(1) for es in ranks]
    for gen in range(3):
        # returns with massive ties (multiples of 0.2 like CartPole's 5-episode means), -0.0, and one clear maximum
        fit = torch.round(torch.rand(n, device="cuda", generator=g) * 200.0) * 0.2
        fit[gen] = -0.0
        fit[(n * 3) // 4 + gen] = 1000.0 + gen
        new = [ref.empty(P) for _ in range(3)]
        ref.openai_generation(fit, 7, gen, 0.05, 0.3, 0.04, state, new, 0.29, gen + 1, 0, n, theta_next=theta_ref, best=best_ref)
        torch.cuda.synchronize()
        for r, es in enumerate(ranks):
            with torch.cuda.stream(streams[r]):
                es.openai_generation(fit, 7, gen, 0.05, 0.3, 0.04, state, outs[r], 0.29, gen + 1, r * per, thetas[r].shape[0],
                                     theta_next=thetas[r], best=bests[r], comm=es, per_rank=per, world=world)
        torch.cuda.synchronize()
        for r, es in enumerate(ranks):
            assert es.comm_p2p_status() == 0, (gen, r, "an exchange timed out")
            for a, b, what in zip(outs[r], new, ("mu", "m", "v")):
                assert torch.equal(a.view(torch.int32), b.view(torch.int32)), (gen, r, what)
            assert torch.equal(bests[r].view(torch.int32), best_ref.view(torch.int32)), (gen, r, float(bests[r]), float(best_ref))
            lo = r * per
            assert torch.equal(thetas[r].view(torch.int32), theta_ref[lo:lo + thetas[r].shape[0]].view(torch.int32)), (gen, r, "theta")
        assert float(best_ref) == 1000.0 + gen
        state = new
    # the all-gather itself in both forms (sequence words / granules), 5000 floats per rank = two slices of the grid
    for gran in (0, 1):
        locals_, outs_ = [], []
        for r, es in enumerate(ranks):
            es.set_tuning("comm_granule_allgather", gran)
            with torch.cuda.stream(streams[r]):
                loc = torch.arange(5000, device="cuda", dtype=torch.float32) + 10000.0 * r + gran
                locals_.append(loc)
                outs_.append(es.allgather_fitness(loc))
        torch.cuda.synchronize()
        want = torch.cat(locals_)
        for r, es in enumerate(ranks):
            assert es.comm_p2p_status() == 0 and torch.equal(outs_[r], want), (gran, r)
    for es in ranks:
        es.comm_p2p_detach()
    print("ok", world, per, n)
""")


@pytest.mark.parametrize("world,per,missing,gru,granules",
                         [(8, 4096, 0, 0, 1), (8, 8192, 0, 0, 1), (16, 4096, 5, 0, 1), (2, 1024, 1, 0, 1), (4, 2048, 0, 1, 1),
                          (8, 1024, 7, 0, 1), (4, 4096, 3, 0, 0), (2, 1024, 0, 1, 0)],
                         ids=["8x4096_weak8", "8x8192_c4", "16x4096_ragged", "2x1024_counting_rank_ragged", "4x2048_gru_P6562",
                              "8x1024_counting_rank_ragged", "4x4096_ragged_float_allgather", "2x1024_gru_float_allgather"])
def test_sharded_tail_equals_replicated_tail_bitwise(tmp_path, world, per, missing, gru, granules):
    script = tmp_path / "tail.py"
    script.write_text(WORKER % (ROOT, SRC))
    cmd = ["timeout", "-k", "10", "300", sys.executable, str(script), str(world), str(per), str(missing), str(gru), str(granules)]
    run = subprocess.run(cmd, capture_output=True, text=True, timeout=400)
    assert run.returncode == 0, run.stdout[-3000:] + run.stderr[-3000:]
    assert run.stdout.strip().endswith(f"ok {world} {per} {world * per - missing}")
