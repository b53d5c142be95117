"""The fitness all-gather of the C ABI (ses_comm_* / ses_allgather_fitness on the handle's stream): RCCL, and the
peer-store transport's argument checks (its data path runs in tests/test_gpu_multirank.py, ranks sharing the GPU).

On the 1-GPU test box RCCL can only form a one-rank communicator (it refuses two ranks on one device), which still
runs the library's whole RCCL path -- dlopen, ncclGetUniqueId, ncclCommInitRank, ncclAllGather on the handle's
stream, ncclCommDestroy.  With two or more GPUs visible the last test runs the sharded generation loop over two real
ranks, once per transport, and compares it bit for bit with one rank."""
import ctypes
import os
import socket
import subprocess
import sys
import textwrap

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "simple-es_amd")


def test_single_rank_rccl_communicator_roundtrip():
    from ses import HipES, SesError
    es = HipES(None, 4, 2, True, False)
    assert es.comm_info()[1] == 0                                    # no communicator yet
    shard = torch.rand(4096, device="cuda")
    with pytest.raises(SesError, match="no communicator"):
        es.allgather_fitness(shard)
    uid = HipES.comm_unique_id()
    assert len(uid) == 128 and any(uid)
    with pytest.raises(SesError):
        es.comm_init(1, 1, uid)                                      # rank outside [0, world)
    es.comm_init(0, 1, uid)
    rank, world, version = es.comm_info()
    assert (rank, world) == (0, 1) and version >= 20000
    with pytest.raises(SesError, match="already has a communicator"):
        es.comm_init(0, 1, uid)
    out = es.allgather_fitness(shard)
    es.sync()
    assert torch.equal(out, shard)
    for n in (1, 7, 513, 65536):                                     # C2 shard = 512..4096 floats, C4 total = 65536
        s = torch.rand(n, device="cuda")
        assert torch.equal(es.allgather_fitness(s), s)
    es.comm_destroy()
    assert es.comm_info()[1] == 0
    es.close()


def test_exclusive_streams_come_and_go():
    """ses_stream_create_exclusive / ses_stream_destroy: a handle works on such a stream, and a rig can give the stream back
    (streams left alive end a rocprofv3 run in SIGSEGV at process exit: NOTES.md, round 6)."""
    from ses import HipES, destroy_stream, exclusive_stream
    for _ in range(3):
        s = exclusive_stream()
        es = HipES("CartPole-v1", 4, 2, True, False, max_step=20, eval_ep_num=2, stream=s)
        with torch.cuda.stream(s):
            theta = torch.zeros(64, es.P, device="cuda")
            init = torch.full((es.E, es.init_dim), 0.01, device="cuda")
            fit = es.rollout(theta, init)
        s.synchronize()
        assert fit.shape == (64,) and float(fit.min()) >= 1.0 and float(fit.max()) <= 20.0
        es.close()
        destroy_stream(s)


def test_raw_abi_comm_error_paths():
    from ses import _lib
    lib = _lib.load()
    cfg = _lib.SesConfig(-1, 4, 2, 1, 0, 0, 500, 5, 0, 0, 1, 0)
    h = ctypes.c_void_p()
    assert lib.ses_create(ctypes.byref(cfg), None, ctypes.byref(h)) == 0
    buf = ctypes.create_string_buffer(128)
    assert lib.ses_comm_unique_id(None) == -1
    assert lib.ses_comm_init(h, 0, 0, buf) == -1                     # world < 1
    assert lib.ses_comm_init(None, 0, 1, buf) == -1
    x = torch.zeros(8, device="cuda")
    assert lib.ses_allgather_fitness(h, ctypes.c_void_p(x.data_ptr()), 8, ctypes.c_void_p(x.data_ptr())) == -1
    assert b"no communicator" in lib.ses_last_error()
    # peer-store transport: argument checks, life cycle of a mailbox nobody else maps
    hb = ctypes.create_string_buffer(_lib.COMM_P2P_HANDLE_BYTES)
    assert lib.ses_comm_p2p_export(h, 0, 1, 4096, hb) == -1           # world < 2
    assert lib.ses_comm_p2p_export(h, 2, 2, 4096, hb) == -1           # rank outside the world
    assert lib.ses_comm_p2p_export(h, 0, 17, 4096, hb) == -1          # more ranks than a node has GPUs
    assert lib.ses_comm_p2p_export(h, 0, 2, 0, hb) == -1
    assert lib.ses_comm_p2p_attach(h, hb) == -1                       # nothing exported yet
    w, m, x_ = ctypes.c_int32(-1), ctypes.c_int32(-1), ctypes.c_int32(-1)
    assert lib.ses_comm_p2p_info(h, ctypes.byref(w), ctypes.byref(m), ctypes.byref(x_)) == 0 and w.value == 0
    assert lib.ses_comm_p2p_export(h, 0, 2, 4097, hb) == 0 and any(hb.raw)
    assert lib.ses_comm_p2p_export(h, 0, 2, 4096, hb) == -1           # one mailbox per handle
    assert lib.ses_comm_p2p_info(h, ctypes.byref(w), None, None) == 0 and w.value == 0    # exported, not attached
    assert lib.ses_allgather_fitness(h, ctypes.c_void_p(x.data_ptr()), 8, ctypes.c_void_p(x.data_ptr())) == -1
    assert lib.ses_comm_p2p_detach(h) == 0 and lib.ses_comm_p2p_detach(h) == 0
    assert lib.ses_destroy(h) == 0


SHARED = textwrap.dedent("""
    import os, sys, torch
    sys.path[:0] = [%r, %r]
    import torch.distributed as dist
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    from ses import HipES
    from ses import parallel
    a = HipES("CartPole-v1", 4, 2, True, False)
    b = HipES("CartPole-v1", 4, 2, True, True)
    assert parallel.attach_comm(a, allow_single=True) and parallel.attach_comm(b, allow_single=True)
    assert a._comm_owner is b._comm_owner and a._comm_owner is not a          # one communicator per process, on its own handle
    assert parallel.comm_info(a)[:2] == (0, 1) and a.comm_info()[1] == 0
    x = torch.rand(4096, device="cuda")
    assert torch.equal(a._comm_owner.allgather_fitness(x), x)
    a.close(); b.close()                                                       # the owner outlives the handles that use it
    assert torch.equal(parallel._COMM[(0, 1)].allgather_fitness(x), x)
    dist.destroy_process_group()
    print("shared-communicator ok")
""")


def test_one_shared_communicator_per_process(tmp_path):
    """attach_comm: the RCCL communicator lives on a dedicated handle shared by every loop of the process (a 1-rank nccl
    process group stands in for the multi-GPU launch on this 1-GPU box)."""
    script = tmp_path / "s.py"
    script.write_text(SHARED % (ROOT, SRC))
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = {**os.environ, "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port)}
    out = subprocess.run([sys.executable, str(script)], capture_output=True, text=True, timeout=600, env=env)
    assert out.returncode == 0 and "shared-communicator ok" in out.stdout, out.stdout[-2000:] + out.stderr[-3000:]


WORKER = textwrap.dedent("""
    import contextlib, io, os, sys
    import numpy as np, torch
    sys.path[:0] = [%r, %r]
    out_dir, world = sys.argv[1], int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1:
        import torch.distributed as dist
        local = int(os.environ["LOCAL_RANK"])
        torch.cuda.set_device(local)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    import builder
    os.chdir(out_dir)
    for name, n in (("openai_es", 203), ("simple_evolution", 96), ("simple_genetic", 120)):
        cfg = {"env": {"name": "CartPole-v1", "max_step": 200, "pomdp": False, "seed": 3},
               "network": {"name": "gym_model", "num_state": 4, "num_action": 2, "discrete_action": True, "gru": False},
               "strategy": {"name": name, "init_sigma": 0.5, "sigma_decay": 0.99, "learning_rate": 0.05,
                            "elite_num": 8, "offspring_num": n, "seed": 5}}
        loop = builder.build_loop(cfg, 4, 1, 3, False, 10 ** 9)
        from ses.parallel import comm_info, comm_transport
        assert comm_info(loop.dev)[1] == (world if world > 1 else 0)
        if world > 1:
            assert comm_transport(loop.dev) == ("rccl" if os.environ.get("SES_COMM_P2P") == "0" else "p2p-store")
        fits = []
        orig = loop.rollout
        loop.rollout = lambda pop, _o=orig: (fits.append(_o(pop).cpu().numpy().copy()) or torch.from_numpy(fits[-1]).cuda())
        with contextlib.redirect_stdout(io.StringIO()):
            loop.run()
        rank = int(os.environ.get("RANK", "0"))
        elite = loop.offspring_strategy.get_elite_model().flat()
        np.savez(os.path.join(out_dir, f"{name}_w{world}_r{rank}.npz"), fits=np.stack(fits), elite=elite,
                 best=np.array([b for b, _ in loop.history]))
    if world > 1:
        dist.destroy_process_group()
""")


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs (RCCL refuses two ranks on one device)")
@pytest.mark.parametrize("p2p", ["1", "0"], ids=["peer_stores_over_xgmi", "rccl"])
def test_two_gpus_equal_one_gpu_bitwise(tmp_path, p2p):
    script = tmp_path / "w.py"
    script.write_text(WORKER % (ROOT, SRC))
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    one = subprocess.run([sys.executable, str(script), str(tmp_path)], capture_output=True, text=True, timeout=600)
    assert one.returncode == 0, one.stdout + one.stderr
    two = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                          "--master-addr", "127.0.0.1", "--master-port", str(port), str(script), str(tmp_path)],
                         capture_output=True, text=True, timeout=900, env={**os.environ, "SES_COMM_P2P": p2p})
    assert two.returncode == 0, two.stdout + two.stderr
    for name in ("openai_es", "simple_evolution", "simple_genetic"):
        ref = np.load(tmp_path / f"{name}_w1_r0.npz")
        for r in (0, 1):
            got = np.load(tmp_path / f"{name}_w2_r{r}.npz")
            assert np.array_equal(got["fits"].view(np.uint32), ref["fits"].view(np.uint32)), (name, r, "fitness")
            assert np.array_equal(got["elite"].view(np.uint32), ref["elite"].view(np.uint32)), (name, r, "elite")
            assert np.array_equal(got["best"], ref["best"])


# ---------------------------------------------------------------------------------------------------------------------------------
# Real devices, worlds of 2 / 4 / 8, the three population shapes bench.py times at N GPUs.  These tests only
# wake up on a multi-GPU node (the pool's boxes have one GPU and a guard of six GPU processes): there they are the first thing that
# says whether the replacement of `Pool.map` (loop.py:66-79) returns every result, in order, on real xGMI links -- on both transports.
SCALE_WORKER = textwrap.dedent("""
    import os, sys
    import numpy as np, torch
    sys.path[:0] = [%r, %r]
    out_dir, per_rank_list, world_run = sys.argv[1], [int(x) for x in sys.argv[2].split(",")], int(sys.argv[3])
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    rig = os.environ.get("SES_TEST_BACKEND", "nccl") == "gloo"        # the ranks share GPU 0 (1-GPU boxes): gloo control plane
    if world > 1:
        import torch.distributed as dist
        local = 0 if rig else int(os.environ["LOCAL_RANK"])
        torch.cuda.set_device(local)
        if rig:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    import builder
    from ses.parallel import comm_info, comm_transport
    os.chdir(out_dir)
    for per in per_rank_list:
        n = per * world_run                     # the single-rank reference run computes the population of the `world_run`-rank job
        cfg = {"env": {"name": "CartPole-v1", "max_step": 60, "pomdp": False, "seed": 11, "shared_init": True, "fixed_length": True},
               "network": {"name": "gym_model", "num_state": 4, "num_action": 2, "discrete_action": True, "gru": False},
               "strategy": {"name": "openai_es", "init_sigma": 0.4, "sigma_decay": 0.995, "learning_rate": 0.05,
                            "offspring_num": n, "seed": 2}}
        loop = builder.build_loop(cfg, 0, 1, 5, False, 10 ** 9)
        strat = loop.offspring_strategy
        pop = strat.init_offspring(loop.network, loop.env.get_agent_ids())
        if world > 1:
            want = "rccl" if os.environ.get("SES_COMM_P2P") == "0" else "p2p-store"
            assert rig or comm_info(loop.dev)[1] == world, comm_info(loop.dev)    # an RCCL communicator over ALL the ranks
            assert comm_transport(loop.dev, pop.shard.per_rank) == want, comm_transport(loop.dev, pop.shard.per_rank)
        # (1) the per-generation calls: every rank's all-gathered fitness vector, two generations
        fits = []
        for g in range(2):
            fit = loop.rollout(pop)
            fits.append(fit.cpu().numpy().copy())
            pop, best, sigma = strat.evaluate(fit)
        # (2) the call bench.py times: four generations through ESLoop.generations (on a library transport the device-side loop,
        # both exchanges inside the kernels around them)
        pop = loop.generations(pop, 4)
        torch.cuda.synchronize()
        sh = pop.shard
        np.savez(os.path.join(out_dir, f"n{n}_w{world}_r{rank}.npz"), fits=np.stack(fits), mu=strat.mu_model.cpu().numpy(),
                 m=strat.optimizer.m.cpu().numpy(), v=strat.optimizer.v.cpu().numpy(), theta=pop.theta.cpu().numpy(),
                 first=np.array(sh.first), n_local=np.array(sh.n_local), device_loop=np.array(bool(loop.device_side_loop)))
        del loop, strat, pop
    if world > 1:
        dist.destroy_process_group()
""")


def _bench_shapes_many_against_one(tmp_path, world, p2p, rig):
    script = tmp_path / "scale.py"
    script.write_text(SCALE_WORKER % (ROOT, SRC))
    pers = [4096 // world, 4096, 8192]
    arg = ",".join(str(p) for p in pers)
    one = subprocess.run([sys.executable, str(script), str(tmp_path), arg, str(world)], capture_output=True, text=True, timeout=900)
    assert one.returncode == 0, one.stdout[-2000:] + one.stderr[-4000:]
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    many = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world),
                           "--master-addr", "127.0.0.1", "--master-port", str(port), str(script), str(tmp_path), arg, str(world)],
                          capture_output=True, text=True, timeout=900,
                          env={**os.environ, "SES_COMM_P2P": p2p, "HSA_ENABLE_IPC_MODE_LEGACY": "0",
                               "SES_TEST_BACKEND": "gloo" if rig else "nccl"})
    assert many.returncode == 0, many.stdout[-2000:] + many.stderr[-4000:]
    for per in pers:
        n = per * world
        ref = np.load(tmp_path / f"n{n}_w1_r0.npz")
        assert ref["fits"].std() > 0 and np.abs(ref["mu"]).sum() > 0
        for r in range(world):
            got = np.load(tmp_path / f"n{n}_w{world}_r{r}.npz")
            for key in ("fits", "mu", "m", "v"):
                assert np.array_equal(got[key].view(np.uint32), ref[key].view(np.uint32)), (n, r, key)
            lo, k = int(got["first"]), int(got["n_local"])
            assert (lo, k) == (r * per, per)
            assert np.array_equal(got["theta"].view(np.uint32), ref["theta"][lo:lo + k].view(np.uint32)), (n, r, "next population")
            assert bool(got["device_loop"]), "a library transport carries the shards: the timed path is the device-side loop"


@pytest.mark.parametrize("p2p", ["1", "0"], ids=["peer_stores_over_xgmi", "rccl"])
@pytest.mark.parametrize("world", [2, 4, 8])
def test_real_gpus_equal_one_gpu_bitwise_at_the_bench_shapes(tmp_path, world, p2p):
    """world GPUs against one, bit for bit, at the three jobs bench.py times on `world` GPUs: the strong job of record (4096
    offspring in total: 8 x 512), the weak job (4096 per GPU) and C4's layout (8192 per GPU: 8 x 8192 = 65 536)."""
    if torch.cuda.device_count() < world:
        pytest.skip(f"needs {world} GPUs (RCCL refuses two ranks on one device), this box shows {torch.cuda.device_count()}")
    _bench_shapes_many_against_one(tmp_path, world, p2p, rig=False)


@pytest.mark.parametrize("world", [2, 4])
def test_the_bench_shapes_on_the_rig_whose_ranks_share_the_gpu(tmp_path, world):
    """The same worker and the same comparisons with the ranks SHARING this box's GPU (gloo control plane, peer stores through
    hipIpc): what a 1-GPU box can say about the test above -- that the test itself is sound -- before a node runs it."""
    _bench_shapes_many_against_one(tmp_path, world, "1", rig=True)
