"""CPU tests of the host-side logic: population sharding arithmetic, the world_size-2 fitness all-gather
over gloo (the N>1 path of bench.py / ESLoop, RCCL on the GPU box), config surface of builder / run_es."""
import os
import subprocess
import sys
import textwrap

import json
import socket

import pytest
import yaml

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "simple-es_amd")


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_shard_partition_covers_population_exactly():
    from ses.parallel import Shard
    import torch.distributed as dist
    assert not dist.is_initialized()
    s = Shard(4097)
    assert (s.world, s.rank, s.first, s.n_local) == (1, 0, 0, 4097)
    # emulate W ranks without a process group
    for n in (1, 2, 7, 96, 97, 4096, 65536):
        for w in (1, 2, 3, 4, 8):
            per = -(-n // w)
            rows = []
            for r in range(w):
                first = min(r * per, n)
                rows += list(range(first, first + max(0, min(per, n - first))))
            assert rows == list(range(n))


WORKER = textwrap.dedent("""
    import sys, torch, torch.distributed as dist
    sys.path.insert(0, %r)
    dist.init_process_group("gloo")
    from ses.parallel import Shard
    for n in (10, 7, 1, 2, 4096):
        sh = Shard(n)
        local = torch.arange(sh.first, sh.first + sh.n_local, dtype=torch.float32) * 0.5
        out = sh.allgather_fitness(local)
        assert out.shape == (n,) and out.tolist() == [0.5 * i for i in range(n)], (n, out)
        sh.barrier()
    open(sys.argv[1] + "/rank" + str(dist.get_rank()) + ".ok", "w").write("ok")
    dist.destroy_process_group()
""")


def test_fitness_allgather_world_size_2_gloo(tmp_path):
    script = tmp_path / "w.py"
    script.write_text(WORKER % SRC)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), str(script), str(tmp_path)]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=240)
    assert out.returncode == 0, out.stdout + out.stderr
    assert (tmp_path / "rank0.ok").exists() and (tmp_path / "rank1.ok").exists()


def test_fitness_allgather_world_size_8_gloo(tmp_path):
    """The 8-rank layout of BASELINE configs[3] on the CPU (gloo): 65 536 rows = 8 x 8192, and the ragged cases
    (last ranks padded with -inf, ranks that own nothing) -- the shard arithmetic of ses/parallel.py, no GPU."""
    script = tmp_path / "w8.py"
    script.write_text(WORKER.replace("(10, 7, 1, 2, 4096)", "(65536, 4097, 10, 3)") % SRC)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "8",
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), str(script), str(tmp_path)]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=480)
    assert out.returncode == 0, out.stdout + out.stderr
    assert all((tmp_path / f"rank{r}.ok").exists() for r in range(8))


AGREE_WORKER = textwrap.dedent("""
    import sys, torch, torch.distributed as dist
    sys.path.insert(0, %r)
    dist.init_process_group("gloo")
    from ses.parallel import all_ranks, attach_comm
    rank = dist.get_rank()
    dev = torch.device("cpu")
    # one rank that cannot take the device-side loop (a hook on its loop object, SES_BATCH_GENERATIONS=0 in its environment,
    # no transport) switches every rank to the per-generation path; everybody able -> everybody takes it
    assert all_ranks(True, dev) is True
    assert all_ranks(rank != 1, dev) is False
    assert all_ranks(False, dev) is False
    class Handle:                      # attach_comm(create=False) binds to nothing and starts no rendezvous when this process has no owner yet
        pass
    h = Handle()
    if rank == 0:                      # called on ONE rank only: must return at once
        assert attach_comm(h, create=False) is False and h._comm_owner is None
    dist.barrier()
    # solo(): ONE rank steps out of the world to recompute what the sharded run produced (bench.py's shard_check): inside the
    # block it owns every row, collective helpers answer locally, attach_comm is a no-op; outside nothing has changed
    from ses.parallel import Shard, solo, world_size
    if rank == 0:
        with solo():
            sh = Shard(4096)
            assert (sh.world, sh.rank, sh.first, sh.n_local) == (1, 0, 0, 4096) and world_size() == 1
            assert all_ranks(False, dev) is False and all_ranks(True, dev) is True      # no collective: rank 1 is not here
            assert attach_comm(h) is False
            x = torch.arange(5.0)
            assert sh.allgather_fitness(x) is x
    sh = Shard(4096)
    assert (sh.world, sh.rank, sh.first, sh.n_local) == (2, rank, 2048 * rank, 2048) and world_size() == 2
    dist.barrier()
    open(sys.argv[1] + "/agree" + str(rank) + ".ok", "w").write("ok")
    dist.destroy_process_group()
""")


def test_the_choice_of_the_loop_is_collective_world_size_2_gloo(tmp_path):
    """ESLoop._generation_batch takes the MIN of the ranks' eligibility over the control plane (ses/parallel.py::all_ranks): the
    device-side loop and the per-generation path issue different exchanges, so ranks must never decide alone; and a strategy built
    on a subset of the ranks (attach_comm(create=False)) never starts a rendezvous."""
    script = tmp_path / "agree.py"
    script.write_text(AGREE_WORKER % SRC)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), str(script), str(tmp_path)]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=240)
    assert out.returncode == 0, out.stdout + out.stderr
    assert (tmp_path / "agree0.ok").exists() and (tmp_path / "agree1.ok").exists()


def test_bench_spawns_its_own_ranks(tmp_path):
    """`python bench.py --gpus 2` without a launcher: the parent starts torch.distributed.run as a child, relays
    rank 0's JSON line and returns the child's exit code (here the ranks only rendezvous: no GPU in this container)."""
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--rendezvous-only"],
                         capture_output=True, text=True, timeout=240, cwd=str(tmp_path),
                         env={k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")})
    assert out.returncode == 0, out.stdout + out.stderr
    line = [ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1]
    assert json.loads(line) == {"rendezvous": "ok", "world": 2, "ranks": [0, 1]}


@pytest.mark.skipif(__import__("torch").cuda.device_count() >= 2, reason="needs a box with fewer than 2 GPUs")
def test_bench_fails_loudly_when_gpus_are_missing(tmp_path):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2"],
                         capture_output=True, text=True, timeout=120, cwd=str(tmp_path),
                         env={k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")})
    assert out.returncode == 3, out.stdout + out.stderr
    assert "needs 2 visible GPUs" in out.stderr and not out.stdout.strip()


def test_config_surface_matches_reference_yaml_keys():
    cfg = yaml.load(open(os.path.join(SRC, "conf", "cartpole.yaml")), Loader=yaml.FullLoader)
    assert set(cfg) == {"env", "network", "strategy"}
    assert set(cfg["env"]) == {"name", "max_step", "pomdp"}
    assert set(cfg["network"]) == {"name", "num_state", "num_action", "discrete_action", "gru"}
    assert set(cfg["strategy"]) == {"name", "init_sigma", "sigma_decay", "elite_num", "offspring_num"}


def test_cli_flags_match_reference():
    out = subprocess.run([sys.executable, os.path.join(SRC, "run_es.py"), "--help"], capture_output=True, text=True,
                         cwd=SRC, timeout=120)
    assert out.returncode == 0, out.stderr
    for flag in ("--cfg-path", "--seed", "--process-num", "--generation-num", "--eval-ep-num", "--log",
                 "--save-model-period"):
        assert flag in out.stdout


def test_network_container_layout_and_checkpoint_keys():
    import numpy as np
    from networks.neural_network import GymEnvModel
    m = GymEnvModel(4, 2, True, True)
    assert list(m.state_dict()) == ["fc1.weight", "fc1.bias", "gru.weight_ih_l0", "gru.weight_hh_l0",
                                    "gru.bias_ih_l0", "gru.bias_hh_l0", "fc2.weight", "fc2.bias"]
    assert m.param_count() == 6562
    vec = np.arange(6562, dtype=np.float32)
    m.load_flat(vec)
    assert np.array_equal(m.flat(), vec)
    assert np.array_equal(m.get_param_list()[0], vec[:128].reshape(32, 4))          # fc1.weight is row-major (32,S)
    m.zero_init()
    assert not m.flat().any()
    views = m.get_param_list()
    views[1] += 1.0                                                                 # views edit the module in place
    assert m.fc1.bias.sum().item() == 32.0


@pytest.mark.skipif(__import__("torch").cuda.is_available(), reason="needs a box without a GPU")
def test_strategies_refuse_to_run_without_gpu():
    from learning_strategies.evolution.offspring_strategies import openai_es
    from networks.neural_network import GymEnvModel
    from ses import SesError
    with pytest.raises(SesError):
        openai_es(0.1, 0.999, 0.05, 16).init_offspring(GymEnvModel(4, 2, True, False), ["0"])


def test_sweep_files_keep_reference_format_and_points_are_reproducible():
    import random
    sys.path.insert(0, SRC)
    import sweep_main
    for name in os.listdir(os.path.join(SRC, "sweep_config")):
        spec = yaml.load(open(os.path.join(SRC, "sweep_config", name)), Loader=yaml.FullLoader)
        assert {"program", "method", "metric", "parameters"} <= set(spec), name
        assert spec["metric"]["name"] == "ep5_mean_reward"
        for key, val in spec["parameters"].items():
            assert set(val) in ({"value"}, {"values"}, {"min", "max"}), (name, key)
            assert "--" + key in sweep_main_help(), key
    spec = yaml.load(open(os.path.join(SRC, "sweep_config", "cartpole_genetic_grid.yaml")), Loader=yaml.FullLoader)
    grid = list(sweep_main.trial_points(spec, None, random.Random(0)))
    assert len(grid) == 6 and {(p["init_sigma"], p["elite_num"]) for p in grid} == {(s, k) for s in (0.5, 1.0, 2.0) for k in (2, 4)}
    assert all(p["cfg_path"] == "conf/cartpole.yaml" and p["generation_num"] == 20 for p in grid)
    # the reference ships two sweep files (sweep_config/bipedal_genetic.yaml, lunarlander_openaies.yaml): both are here
    assert {"bipedal_genetic.yaml", "lunarlander_openaies.yaml"} <= set(os.listdir(os.path.join(SRC, "sweep_config")))
    spec = yaml.load(open(os.path.join(SRC, "sweep_config", "bipedal_genetic.yaml")), Loader=yaml.FullLoader)
    pts = list(sweep_main.trial_points(spec, 7, random.Random(1)))
    assert len(pts) == 7 and pts == list(sweep_main.trial_points(spec, 7, random.Random(1)))
    assert all(p["cfg_path"] == "conf/bipedalwalker.yaml" and p["generation_num"] == 300 and p["offspring_num"] == 96 and
               p["elite_num"] in (5, 10, 15, 20) and 0.05 <= p["init_sigma"] <= 2.0 and p["sigma_decay"] in (0.999, 0.9999)
               for p in pts)
    spec = yaml.load(open(os.path.join(SRC, "sweep_config", "cartpole_openaies.yaml")), Loader=yaml.FullLoader)
    a = list(sweep_main.trial_points(spec, 5, random.Random(3)))
    b = list(sweep_main.trial_points(spec, 5, random.Random(3)))
    assert a == b and len(a) == 5
    assert all(0.001 <= p["learning_rate"] <= 0.2 and 0.05 <= p["init_sigma"] <= 1.0 and
               p["sigma_decay"] in (0.999, 0.9999) for p in a)


_HELP = {}


def sweep_main_help():
    if "out" not in _HELP:
        out = subprocess.run([sys.executable, os.path.join(SRC, "sweep_main.py"), "--help"], capture_output=True,
                             text=True, cwd=SRC, timeout=120)
        assert out.returncode == 0, out.stderr
        _HELP["out"] = out.stdout
    return _HELP["out"]


def test_spread_collision_threshold_is_where_the_root_reaches_the_contact_distance():
    """csrc/ses_spread.h counts collisions on the SQUARED distance: s < SP_DIST_MIN_SQ must say what sqrtf(s) < 0.3f says
    (oracle/ses_oracle.c::spread_step) for every float s -- true when the constant is the smallest float whose correctly
    rounded root reaches 0.3f, sqrtf being monotonic."""
    import re
    import numpy as np
    text = open(os.path.join(SRC, "csrc", "ses_spread.h")).read()
    t = np.float32(float.fromhex(re.search(r"SP_DIST_MIN_SQ = (0x[0-9a-fA-F.]+p[-+]?\d+)f;", text).group(1)))
    d = np.float32(float(re.search(r"SP_DIST_MIN = ([0-9.]+)f;", text).group(1)))
    assert d == np.float32(0.3)
    assert np.sqrt(t) >= d and np.sqrt(np.nextafter(t, np.float32(0))) < d
    # and the roots around it are ordered as their arguments (a window of 2^16 floats on either side)
    s = (t.view(np.uint32) + np.arange(-65536, 65536, dtype=np.int64)).astype(np.uint32).view(np.float32)
    r = np.sqrt(s)
    assert np.all(np.diff(r) >= 0) and np.array_equal(r < d, s < t)


def test_bench_timing_helpers_on_the_cpu():
    """bench.py's host-side bookkeeping (no GPU): the timed blocks repeat EXACTLY `steps` generations per block until they cover
    --min-timed-seconds (never fewer than --blocks), the summary is the median block, the leg timers add up wall seconds."""
    import importlib.util
    import time
    spec = importlib.util.spec_from_file_location("bench_helpers", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)

    class FakeJob:
        n_global, world, E, T = 4096, 1, 5, 500
        calls = []

        def generations(self, k):
            self.calls.append(k)
            time.sleep(0.004)

        def steps_per_generation(self):
            return self.n_global * self.E * self.T

    job = FakeJob()
    times = bench.timed_blocks(job, 7, 3, lambda: None, None, 1, min_seconds=0.05)
    assert len(times) >= 3 and sum(times) >= 0.05 and set(job.calls) == {7} and len(job.calls) == len(times)
    few = bench.timed_blocks(job, 7, 3, lambda: None, None, 1)               # no minimum: exactly --blocks blocks
    assert len(few) == 3
    rec = bench.summarise(job, 7, times)
    assert rec["blocks"] == len(times) and rec["steps"] == 7 and rec["offspring_total"] == 4096
    assert abs(rec["value"] - job.steps_per_generation() / (rec["ms_per_step"] * 1e-3)) < 1e-3 * rec["value"]
    assert rec["ms_per_step_min"] <= rec["ms_per_step"] <= rec["ms_per_step_max"] and abs(rec["timed_seconds"] - sum(times)) < 1e-9
    legs = bench.Legs()
    with legs.leg("a", gpu=False):
        time.sleep(0.01)
    legs.begin("b", gpu=False)
    legs.end()
    with legs.leg("a", gpu=False):
        pass
    assert legs.wall["a"] >= 0.01 and "b" in legs.wall and legs.gpu == {}
