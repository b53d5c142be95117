"""Several ranks sharing the one GPU of the test box (gloo rendezvous; RCCL refuses duplicate devices): the sharded
generation loop must give every rank the identical fitness vector and parent, bit-equal to a single-rank run.
This is the population-sharding path that bench.py --gpus N / run_es.py use on a multi-GPU node, and the fitness
all-gather is the library's own peer-store transport (ses_comm_p2p_*: mailboxes mapped across processes with hipIpc --
same mechanism across GPUs over xGMI); with SES_COMM_P2P=0 the torch.distributed fallback carries it."""
import os
import socket
import subprocess
import sys
import textwrap

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "simple-es_amd")

WORKER = textwrap.dedent("""
    import contextlib, io, os, sys
    import numpy as np, torch, yaml
    sys.path[:0] = [%r, %r]
    out_dir, world = sys.argv[1], int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("gloo")
    import builder
    os.chdir(out_dir)
    for name, n in (("openai_es", 203), ("simple_evolution", 96), ("simple_genetic", 120)):
        cfg = {"env": {"name": "CartPole-v1", "max_step": 200, "pomdp": False, "seed": 3},
               "network": {"name": "gym_model", "num_state": 4, "num_action": 2, "discrete_action": True, "gru": False},
               "strategy": {"name": name, "init_sigma": 0.5, "sigma_decay": 0.99, "learning_rate": 0.05,
                            "elite_num": 8, "offspring_num": n, "seed": 5}}
        loop = builder.build_loop(cfg, 4, 1, 3, False, 10 ** 9)
        fits = []
        orig = loop.rollout
        loop.rollout = lambda pop, _o=orig: (fits.append(_o(pop).cpu().numpy().copy()) or torch.from_numpy(fits[-1]).cuda())
        with contextlib.redirect_stdout(io.StringIO()):
            loop.run()
        rank = int(os.environ.get("RANK", "0"))
        from ses.parallel import comm_transport
        open(os.path.join(out_dir, f"{name}_w{world}_r{rank}.transport"), "w").write(comm_transport(loop.dev))
        elite = loop.offspring_strategy.get_elite_model().flat()
        np.savez(os.path.join(out_dir, f"{name}_w{world}_r{rank}.npz"), fits=np.stack(fits), elite=elite,
                 best=np.array([b for b, _ in loop.history]))
    if world > 1:
        dist.destroy_process_group()
""")


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("world,p2p", [(2, True), (4, True), (2, False)], ids=["2_ranks_peer_stores", "4_ranks_peer_stores", "2_ranks_torch_fallback"])
def test_ranks_equal_one_rank_bitwise(tmp_path, world, p2p):
    script = tmp_path / "w.py"
    script.write_text(WORKER % (ROOT, SRC))
    one = subprocess.run([sys.executable, str(script), str(tmp_path)], capture_output=True, text=True, timeout=600)
    assert one.returncode == 0, one.stdout + one.stderr
    env = {**os.environ, "SES_COMM_P2P": "1" if p2p else "0"}
    many = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world),
                           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), str(script), str(tmp_path)],
                          capture_output=True, text=True, timeout=900, env=env)
    assert many.returncode == 0, many.stdout + many.stderr
    for name in ("openai_es", "simple_evolution", "simple_genetic"):
        ref = np.load(tmp_path / f"{name}_w1_r0.npz")
        for r in range(world):
            got = np.load(tmp_path / f"{name}_w{world}_r{r}.npz")
            assert np.array_equal(got["fits"].view(np.uint32), ref["fits"].view(np.uint32)), (name, r, "fitness")
            assert np.array_equal(got["elite"].view(np.uint32), ref["elite"].view(np.uint32)), (name, r, "elite")
            assert np.array_equal(got["best"], ref["best"])
            transport = open(tmp_path / f"{name}_w{world}_r{r}.transport").read()
            assert transport == ("p2p-store" if p2p else "torch"), (name, r, transport, many.stderr[-2000:])
        assert ref["fits"].shape[0] == 4 and ref["fits"].std() > 0
