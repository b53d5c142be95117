"""ses_run_generations -- the C loop ESLoop.run() and bench.py drive -- at the world sizes of the 8-GPU lines of record,
formed in ONE process on the one GPU of the test box: every rank a handle on a stream with a hardware queue of its own
(ses_stream_create_exclusive), the mailboxes attached directly (ses_comm_p2p_attach_local).  Protocol, kernels and layouts
are those of one process per GPU: rollout of the own rows, the fitness exchange fused into the episode-mean kernel and the
rank kernel (granules), the openai_es tail in shard form or replicated, the own rows of the next population.

    8 x  512 = 4096     BASELINE's metric as written at 8 GPUs: shards no 1024-row chunk divides -> replicated tail, the counting
                        rank polls the fitness granules
    8 x 8192 = 65 536   BASELINE configs[3]: shard form of the tail, partials as granules.  On THIS rig with the fitness exchange as a
                        launch of its own ("fused_fitness_exchange" = 0): fused, each rank's sort + search grid is 512 workgroups that
                        spin until the peers' episode means arrive -- eight ranks' worth of them on ONE GPU take every wave slot and
                        the last ranks' rollouts, which they wait for, can never start (the coupling openai_fused_fitness_ok's
                        512-workgroup bound guards against for ONE rank per GPU; across GPUs there is none)
    8 x 2048 = 16 384   the same shard form with BOTH exchanges fused (32 polling workgroups per rank)
    8 x 4096 = 32 768   the weak leg at 8 GPUs, both exchanges fused
    8 x 1024 - 5        ragged last shard, counting rank, shard form
    16 x 512 - 3        the transport's largest world

After k generations every rank's parent, Adam moments, best rewards and rows of the next population equal the ONE-rank run of
the same C loop bit for bit (loop.py:61-104 evaluated once)."""
import os
import subprocess
import sys
import textwrap

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "simple-es_amd")

WORKER = textwrap.dedent("""
    import ctypes, os, sys
    world, per, missing, K = (int(v) for v in sys.argv[1:5])
    tuning = [kv.split("=") for kv in sys.argv[5].split(",")] if len(sys.argv) > 5 and sys.argv[5] else []
    import numpy as np, torch
    sys.path[:0] = [%r, %r]
    from ses import HipES, exclusive_stream, _lib, MODE_EPISODIC
    n, E, T = world * per - missing, 2, 20
    P = 226

    def handle(stream=None):
        return HipES("CartPole-v1", 4, 2, True, False, max_step=T, eval_ep_num=E, stream=stream)

    def state(es, first, n_loc, w, per_rank, comm):
        keep = {"theta": [es.empty(max(n_loc, 1), P), es.empty(max(n_loc, 1), P)],
                "parents": [es.zeros(P), es.empty(P)], "m": [es.zeros(P), es.empty(P)], "v": [es.zeros(P), es.empty(P)],
                "fitness": es.empty(per_rank * w if w > 1 else n), "init": es.empty(1, E, 4),
                "fit_local": torch.full((per_rank,), float("-inf"), device="cuda"),
                "best": torch.full((K,), float("nan")).pin_memory()}
        # generation 0's population: member 0 = mu (the zero network), the others mu + sigma * eps(seed, generation 0, GLOBAL row)
        if n_loc:
            es.perturb(keep["parents"][0], 0.1, 7, 0, first, n_loc, out=keep["theta"][0][:n_loc])
            if first == 0:
                keep["theta"][0][0].zero_()
        st = _lib.SesGenState()
        st.strategy, st.n, st.mode, st.elite_num = _lib.STRATEGY_OPENAI_ES, n, MODE_EPISODIC, 0
        st.shared_init, st.init_width, st.init_lo, st.init_hi = 1, 4, -0.05, 0.05
        st.seed, st.env_seed, st.learning_rate, st.sigma_decay = 7, 3, 0.05, 0.999
        st.sigma = st.pop_sigma = 0.1
        st.pop_gen, st.adam_t, st.cur = 0, 0, 0
        for i in (0, 1):
            st.theta[i], st.parents[i] = keep["theta"][i].data_ptr(), keep["parents"][i].data_ptr()
            st.adam_m[i], st.adam_v[i] = keep["m"][i].data_ptr(), keep["v"][i].data_ptr()
        st.fitness, st.init = keep["fitness"].data_ptr(), keep["init"].data_ptr()
        if w > 1:
            st.world, st.per_rank, st.n_local, st.first_row = w, per_rank, n_loc, first
            st.comm, st.fit_local = comm._h.value, keep["fit_local"].data_ptr()
        return st, keep

    ref = handle()
    st0, keep0 = state(ref, 0, n, 1, n, None)
    ref.run_generations(st0, K, keep0["best"])
    torch.cuda.synchronize()

    streams = [exclusive_stream() for _ in range(world)]
    ranks = [handle(streams[r]) for r in range(world)]
    for r, es in enumerate(ranks):
        es.set_tuning("comm_p2p_timeout_ms", 5000)
        for name, value in tuning:
            es.set_tuning(name, int(value))
        es.comm_p2p_export(r, world, 65536)
    for es in ranks:
        es.comm_p2p_attach_local(ranks)
    sts = []
    for r, es in enumerate(ranks):
        first = min(r * per, n)
        with torch.cuda.stream(streams[r]):
            sts.append(state(es, first, max(0, min(per, n - first)), world, per, es))
    torch.cuda.synchronize()
    # every call only enqueues: rank 0's kernels wait (on the device) for the kernels the later calls enqueue
    for r, es in enumerate(ranks):
        with torch.cuda.stream(streams[r]):
            es.run_generations(sts[r][0], K, sts[r][1]["best"])
    torch.cuda.synchronize()
    cur0 = st0.cur
    assert np.isfinite(keep0["best"].numpy()).all() and keep0["best"].numpy().max() <= T
    fused = ranks[0].comm_p2p_counts()
    for r, es in enumerate(ranks):
        st, keep = sts[r]
        assert es.comm_p2p_status() == 0, (r, "an exchange timed out")
        assert st.cur == cur0 and st.adam_t == K and st.pop_gen == K
        for key in ("parents", "m", "v"):
            assert torch.equal(keep[key][st.cur].view(torch.int32), keep0[key][cur0].view(torch.int32)), (r, key)
        assert torch.equal(keep["best"].view(torch.int32), keep0["best"].view(torch.int32)), (r, keep["best"], keep0["best"])
        first, n_loc = int(st.first_row), int(st.n_local)
        assert torch.equal(keep["theta"][st.cur][:n_loc].view(torch.int32), keep0["theta"][cur0][first:first + n_loc].view(torch.int32)), (r, "theta")
    for es in ranks:
        es.comm_p2p_detach()
    print("ok", world, per, n, "exchanges (all-gather launches, granule exchanges):", fused)
""")


@pytest.mark.parametrize("world,per,missing,tuning",
                         [(8, 512, 0, ""), (8, 8192, 0, "fused_fitness_exchange=0"), (8, 2048, 0, ""), (8, 4096, 0, ""), (8, 1024, 5, ""),
                          (16, 512, 3, ""), (2, 6000, 1, "")],
                         ids=["8x512_metric_as_written", "8x8192_c4", "8x2048_sort_fused", "8x4096_weak", "8x1024_ragged", "16x512_ragged",
                              "2x6000_replicated_sort_allgather_launch"])
def test_device_loop_of_many_ranks_equals_one_rank_bitwise(tmp_path, world, per, missing, tuning):
    script = tmp_path / "loop8.py"
    script.write_text(WORKER % (ROOT, SRC))
    K = 3
    run = subprocess.run(["timeout", "-k", "10", "300", sys.executable, str(script), str(world), str(per), str(missing), str(K), tuning],
                         capture_output=True, text=True, timeout=400)
    assert run.returncode == 0, run.stdout[-3000:] + run.stderr[-3000:]
    last = run.stdout.strip().splitlines()[-1]
    assert last.startswith(f"ok {world} {per} {world * per - missing}"), last
    launches, granules = (int(v) for v in last.split("(")[-1].rstrip(")").split(","))
    if world == 8 and per in (2048, 4096):
        assert (launches, granules) == (0, 2 * K), last          # neither exchange of a generation was a launch of its own
