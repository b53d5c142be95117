#!/usr/bin/env python3
"""bench.py -- env-steps/sec of the simple-es population rollout + fitness loop on MI355X.

    python bench.py --gpus N --steps K --warmup W
    (N > 1: launched by torch.distributed.run, one rank per GPU, RCCL)

Workload (BASELINE.json metric / configs[1]): CartPole-v1, openai_es, MLP policy (P = 226),
4096 offspring PER GPU, eval_ep_num = 5, synthetic fixed-length episodes of 500 steps (termination
masked: every counted env-step is a full policy forward + physics step).  One "step" = one
generation of the hot path, everything on device:
    Philox perturbation -> fused rollout kernel -> fitness all-gather (RCCL, N > 1) ->
    rank-centring -> ES gradient + Adam.
Inputs (mu, Adam moments, initial states) are resident in HBM before the timed region.

Extra legs (rank 0): `roofline` -- the standalone SoA env-step kernel at 2^24 envs against the HBM
roof (52 algorithmic bytes per env-step), timed with HIP events on the launch stream;
`cpu_baseline` -- the reference-structured Python multiprocessing port on the host cores (N=1 only).
"""
import argparse
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "simple-es_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

import torch  # noqa: E402

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6290 measured achievable
BYTES_PER_ENV_STEP = 52        # 7 dword loads + 6 dword stores


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    # 50 + 500 generations = 0.14 s of GPU time: the first few dozen generations after an idle period run ~8 % slower
    # (clock ramp: 0.271 ms per generation at --steps 20 --warmup 3, 0.248 ms at these defaults, 0.251 ms at 2000/200)
    ap.add_argument("--steps", type=int, default=500)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--preroll", type=int, default=-1,
                    help="untimed generations run before the warm-up so that the clocks have ramped whatever --warmup "
                         "is (default: enough to make preroll + warmup = 300, about 75 ms); reported in config")
    ap.add_argument("--offspring-per-gpu", type=int, default=4096)
    ap.add_argument("--eval-ep-num", type=int, default=5)
    ap.add_argument("--max-step", type=int, default=500)
    ap.add_argument("--lanes-per-env", type=int, default=0)
    ap.add_argument("--roofline-envs", type=int, default=1 << 24)
    ap.add_argument("--gru", action="store_true", help="GRU policy on POMDP CartPole instead of the headline MLP workload")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    return ap.parse_args()


def env_step_roofline(es, n_env, launches=20):
    """Average duration of k_env_step_cartpole over `launches` launches, HIP events on the launch stream."""
    from ses import MODE_FIXED_LENGTH
    g = torch.Generator(device="cuda").manual_seed(0)
    x, xd, th, thd, action, ret, status = es.alloc_env_soa(n_env)
    for t in (x, xd, th, thd):
        t.copy_((torch.rand(n_env, device="cuda", generator=g) - 0.5) * 0.1)
    action.copy_((torch.rand(n_env, device="cuda", generator=g) > 0.5).to(torch.int32))
    st = [x, xd, th, thd]
    for _ in range(3):
        es.env_step(*st, action, ret, status, mode=MODE_FIXED_LENGTH)
    torch.cuda.synchronize()
    # `launches` back-to-back launches between two HIP events on the launch stream (torch's current stream
    # IS the handle's stream): the queue never drains, so the quotient is the kernel's own duration
    # (cross-checked against rocprofv3 --kernel-trace in profiles/).
    batches = []
    for _ in range(7):                                         # median batch: a transient slow batch (~8 %) shows up
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)   # about once in ten
        e0.record()
        for _ in range(launches):
            es.env_step(*st, action, ret, status, mode=MODE_FIXED_LENGTH)
        e1.record()
        e1.synchronize()
        batches.append(e0.elapsed_time(e1) * 1e-3 / launches)
    avg = sorted(batches)[len(batches) // 2]
    achieved = BYTES_PER_ENV_STEP * n_env / avg / 1e9
    # this box's streaming ceiling for the same byte count: a plain device-to-device copy (read half, write half)
    half = BYTES_PER_ENV_STEP * n_env // 2
    src, dst = torch.empty(half, dtype=torch.uint8, device="cuda"), torch.empty(half, dtype=torch.uint8, device="cuda")
    dst.copy_(src)
    copies = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            dst.copy_(src)
        e1.record()
        e1.synchronize()
        copies.append(e0.elapsed_time(e1) * 1e-3 / 10)
    copy_gbs = 2 * half / sorted(copies)[len(copies) // 2] / 1e9
    del src, dst
    return {"bound": "hbm", "kernel": "k_env_step_cartpole_v4", "achieved": achieved, "peak": HBM_PEAK_GBS,
            "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": None,
            "n_env": n_env, "avg_launch_us": avg * 1e6, "batch_avg_us": [round(b * 1e6, 2) for b in batches],
            "copy_same_bytes_gbs": copy_gbs, "frac_of_copy": achieved / copy_gbs,
            "env_steps_per_s": n_env / avg,
            "bytes_per_env_step": BYTES_PER_ENV_STEP}


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py --gpus N>1 must be launched with torch.distributed.run (one rank per GPU)")
        args.gpus = world
    torch.cuda.set_device(local_rank % max(torch.cuda.device_count(), 1))
    local_rank = local_rank % max(torch.cuda.device_count(), 1)
    dist = None
    if world > 1:
        import torch.distributed as dist
        backend = os.environ.get("SES_BENCH_BACKEND", "nccl")      # "gloo": test rigs where ranks share one GPU
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)

    from ses import HipES, MODE_FIXED_LENGTH
    from ses.parallel import Shard

    n_local, E, T = args.offspring_per_gpu, args.eval_ep_num, args.max_step
    n_global = n_local * world
    first = rank * n_local
    es = HipES("CartPole-v1", 4, 2, True, args.gru, pomdp=args.gru, max_step=T, eval_ep_num=E, device=local_rank,
               lanes_per_env=args.lanes_per_env)
    lr, sigma0, decay, seed = 0.05, 0.1, 0.999, 0
    mu, m, v = es.zeros(es.P), es.zeros(es.P), es.zeros(es.P)
    init = es.init_states_uniform(seed, 0, 0, 1, shared=True)[0].contiguous()      # [E,4], common random numbers
    theta = es.empty(n_local, es.P)
    fit_local = es.empty(n_local)
    shard = Shard(n_global)                                   # rank r owns rows [r*n_local, (r+1)*n_local)
    assert (shard.first, shard.n_local) == (first, n_local)
    state = {"sigma": sigma0, "t": 0}

    def generation(gen):
        es.perturb(mu, state["sigma"], seed, gen, first, n_local, out=theta)        # row 0 of a real run is mu itself
        es.rollout(theta, init, mode=MODE_FIXED_LENGTH, fitness=fit_local)
        fit_all = shard.allgather_fitness(fit_local)          # RCCL all-gather of N*4 bytes (no-op at world 1)
        _, w = es.rank_center(fit_all)
        state["t"] += 1
        t = state["t"]
        a = lr * math.sqrt(1 - 0.999 ** t) / (1 - 0.99 ** t)
        es.es_update_philox(w, seed, gen, lr, state["sigma"], a, mu, m, v, skip_row0=False)
        state["sigma"] *= decay

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    preroll = args.preroll if args.preroll >= 0 else max(0, 300 - args.warmup)
    for g in range(preroll):                                  # clock ramp; a generation is ~0.25 ms
        generation(10 ** 7 + g)
    mu.zero_(); m.zero_(); v.zero_()                           # the measured run starts from the same state as ever
    state.update(sigma=sigma0, t=0)
    for g in range(args.warmup):
        generation(g)
    barrier()
    t0 = time.perf_counter()
    for g in range(args.warmup, args.warmup + args.steps):
        generation(g)
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([dt], device="cuda", dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())

    steps_per_gen = n_global * E * T
    value = steps_per_gen * args.steps / dt
    result = {
        "metric": "env-steps/sec (whole node), CartPole openai_es pop=4096 at 1/2/4/8 GPUs",
        "value": value, "unit": "env-steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": ("POMDP CartPole-v1 openai_es GRU(4-32-GRU32-2, P=6562)" if args.gru else
                                "CartPole-v1 openai_es MLP(4-32-2, P=226)") + ", fixed-length episodes, termination masked",
                   "offspring_per_gpu": n_local, "offspring_total": n_global, "eval_ep_num": E, "max_step": T,
                   "env_steps_per_generation": steps_per_gen, "noise": "rocRAND philox4x32_10",
                   "preroll_generations": preroll,
                   "parallelism": f"population sharded over {world} GPU(s), fitness all-gather"},
    }
    if rank == 0:
        # per-kernel view of one generation (rank 0, HIP events on the launch stream)
        samples = []
        torch.cuda.synchronize()
        for rep in range(9):                                   # median of 9 back-to-back generations' kernels
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
            ev[0].record()
            es.perturb(mu, state["sigma"], seed, 10 ** 6 + rep, first, n_local, out=theta)
            ev[1].record()
            es.rollout(theta, init, mode=MODE_FIXED_LENGTH, fitness=fit_local)
            ev[2].record()
            ev[2].synchronize()
            samples.append((ev[1].elapsed_time(ev[2]), ev[0].elapsed_time(ev[1])))
        samples.sort()
        roll_ms, perturb_ms = samples[len(samples) // 2]
        result["rollout_kernel"] = {"ms": roll_ms, "perturb_ms": perturb_ms,
                                    "env_steps_per_s_one_gpu": n_local * E * T / (roll_ms * 1e-3),
                                    "bound": "valu-issue/latency (state and weights in VGPRs, no HBM traffic in the loop)"}
        if args.gru and E >= int(os.environ.get("SES_GRU_MFMA_MIN_E", "12")):
            # the GRU rollout runs on v_mfma_f32_16x16x4_f32 from 12 episodes up: 2 fc1 + 96 gate + 8 fc2 tiles per
            # step and 16-episode batch, 2048 flop per tile instruction (padding columns included), fp32 MFMA peak
            # 157.3 TFLOP/s (MI355X_MICROARCH.md, Matrix cores)
            flops = 106 * 2048.0 * n_local * ((E + 15) // 16) * T
            result["rollout_kernel"].update({"bound": "fp32 mfma + valu (serial)", "mfma_tflops": flops / (roll_ms * 1e-3) / 1e12,
                                             "mfma_peak_tflops": 157.3,
                                             "mfma_frac": flops / (roll_ms * 1e-3) / 157.3e12})
        sq = os.path.join(ROOT, "profiles", "r01_sq_rollout.json")
        if os.path.exists(sq) and not args.gru and n_local == 4096 and E == 5 and T == 500 and args.lanes_per_env == 0:
            # VALU issue roofline of the fused kernel: instruction count from the committed SQ counter profile of
            # this same workload, duration measured live above
            prof = json.load(open(sq))
            rate = prof["per_dispatch"]["SQ_INSTS_VALU"] / (roll_ms * 1e-3)
            model = prof.get("issue_cycles_per_step")
            if model:
                # serial-issue estimate: SIMD cycles the loop bodies need if every instruction issued alone at its
                # measured cadence, over the SIMD cycles the kernel had (1024 SIMDs x duration x measured clock)
                need = sum(v["waves"] * v["cycles"] for k, v in model.items() if isinstance(v, dict)) * T
                have = 1024 * roll_ms * 1e-3 * prof["clock_ghz_under_load"] * 1e9
                result["rollout_kernel"]["valu_issue_model_frac"] = need / have
            result["rollout_kernel"].update({"valu_wave_instr_per_s": rate,
                                             "valu_issue_peak_per_s": prof["peak_valu_wave_instr_per_s"],
                                             "valu_issue_frac": rate / prof["peak_valu_wave_instr_per_s"],
                                             "valu_source": "profiles/r01_sq_rollout.json"})
        if not args.no_roofline:
          try:
            result["roofline"] = env_step_roofline(es, args.roofline_envs)
            pmc = os.path.join(ROOT, "profiles", "r01_pmc_env_step.json")
            if os.path.exists(pmc) and args.roofline_envs == (1 << 24):
                # HBM bytes per launch from the committed rocprofv3 PMC passes of this same command
                # (FETCH_SIZE x2 gfx950 correction + WRITE_SIZE, tools/prof_pmc.sh); not collectable in-process
                result["roofline"]["traffic"] = json.load(open(pmc))["traffic_bytes_per_launch"]
                result["roofline"]["traffic_source"] = "profiles/r01_pmc_env_step.json"
          except Exception as exc:                                   # the headline line must still be printed
            result["roofline_error"] = repr(exc)
        if world == 1 and not args.no_cpu_baseline:
            try:
                from oracle import ref_port
                result["cpu_baseline"] = ref_port.time_baseline()
            except Exception as exc:
                result["cpu_baseline_error"] = repr(exc)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(result))


if __name__ == "__main__":
    main()
