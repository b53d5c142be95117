#!/usr/bin/env python3
"""Randomised parity sweep: many small random rollout configurations, HIP path vs the C oracle, bit for bit.
Not part of the test suite (minutes of oracle time); run on a GPU box:  python tools/fuzz_parity.py --cases 300
The oracle is one CPU thread per process: tools/fuzz_round.sh runs several seeds side by side (at most 5 processes may hold
the GPU next to the shell) and adds the tallies up.  Box2D kinds force every kernel of the lander / walker family through
ses_set_tuning (lanes per env 1 ... 64, one / two / four offspring per wave, episode-parallel, lockstep, MFMA), `envstep`
drives the step-wise entries (ses_env_reset / ses_env_step_generic) against the oracle's env objects, `sharded_tail` the shard
form of the openai_es tail on two in-process ranks against the replicated tail."""
import argparse
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "simple-es_amd"))
from oracle import c_oracle as co  # noqa: E402
from ses import HipES  # noqa: E402


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def bits(a):
    a = np.ascontiguousarray(a)
    return a.view(np.uint32 if a.dtype == np.float32 else np.uint64)


_RANK_STREAMS = []


def sharded_tail_case(rng, force=None):
    """One random layout of the shard form of the openai_es tail (ses_openai_generation_sharded) against the replicated tail:
    2 ranks as handles of this process on streams of their own, shards of 1-5 chunks, a ragged last shard, every policy size, both
    rank paths (counting / sort + search) and both ways the chunk partials travel (granules / float all-gather).  Two ranks only:
    an exchange kernel waits for kernels of its peer's stream, so every stream needs a hardware queue of its own
    (ses.exclusive_stream), and five fuzz processes side by side already share the GPU's queue slots (larger worlds:
    tests/test_gpu_sharded_tail.py)."""
    world = 2
    per = 1024 * int(rng.choice([1, 1, 2, 3, 4, 5]))
    n = world * per - int(rng.randint(0, world))
    S, A, gru = [(4, 2, False), (8, 4, False), (24, 4, False), (12, 5, False), (4, 2, True)][int(rng.randint(0, 5))]
    if force:
        per, n, (S, A, gru) = force["per"], force["n"], force["shape"]
    # streams with a hardware queue of their own (ses_stream_create_exclusive), created once per process: an exchange kernel
    # waits for a kernel of its peer's stream, which must not be queued behind it
    from ses import exclusive_stream
    while len(_RANK_STREAMS) < world:
        _RANK_STREAMS.append(exclusive_stream())
    streams = _RANK_STREAMS[:world]
    ranks = [HipES(None, S, A, A in (2, 5), gru, stream=streams[r]) for r in range(world)]
    ref = HipES(None, S, A, A in (2, 5), gru)
    granules = int(rng.randint(0, 2))
    if force:
        granules = force.get("granules", granules)
    for r, es in enumerate(ranks):
        es.set_tuning("comm_p2p_timeout_ms", 20000)
        es.set_tuning("openai_sharded_min_rows", 0)               # the fuzz covers the shard form at every aligned size
        es.set_tuning("openai_granule_exchange", granules)
        es.comm_p2p_export(r, world, 65536)
    for es in ranks:
        es.comm_p2p_attach_local(ranks)
    P = ref.P
    g = torch.Generator(device="cuda").manual_seed(int(rng.randint(0, 2 ** 31 - 1)))
    state = [torch.randn(P, device="cuda", generator=g) * 0.1, torch.randn(P, device="cuda", generator=g) * 0.01,
             torch.rand(P, device="cuda", generator=g) * 0.01]
    ok = ranks[0].openai_sharded_ok(ranks[0], n, per, world)
    ties = float(rng.choice([0.2, 5.0, 1e-4]))
    for gen in range(2):
        fit = torch.round(torch.rand(n, device="cuda", generator=g) * 200.0 / ties) * ties
        if rng.rand() < 0.3:
            fit[int(rng.randint(0, n))] = float("-inf")
        new = [ref.empty(P) for _ in range(3)]
        theta_ref, best_ref = ref.empty(n, P), ref.empty(1)
        sigma, lr, a = float(rng.choice([0.05, 0.5])), 0.05, float(rng.uniform(0.01, 0.06))
        ref.openai_generation(fit, 7, gen, lr, sigma, a, state, new, sigma * 0.99, gen + 1, 0, n, theta_next=theta_ref, best=best_ref)
        torch.cuda.synchronize()
        outs = [[es.empty(P) for _ in range(3)] for es in ranks]
        thetas = [es.empty(min(per, n - r * per), P) for r, es in enumerate(ranks)]
        bests = [es.empty(1) for es in ranks]
        for r, es in enumerate(ranks):
            with torch.cuda.stream(streams[r]):
                es.openai_generation(fit, 7, gen, lr, sigma, a, state, outs[r], sigma * 0.99, gen + 1, r * per, thetas[r].shape[0],
                                     theta_next=thetas[r], best=bests[r], comm=es, per_rank=per, world=world)
        torch.cuda.synchronize()
        for r, es in enumerate(ranks):
            ok = ok and es.comm_p2p_status() == 0
            ok = ok and all(torch.equal(x.view(torch.int32), y.view(torch.int32)) for x, y in zip(outs[r], new))
            ok = ok and torch.equal(bests[r].view(torch.int32), best_ref.view(torch.int32))
            ok = ok and torch.equal(thetas[r].view(torch.int32), theta_ref[r * per:r * per + thetas[r].shape[0]].view(torch.int32))
        state = new
    for es in ranks:
        es.comm_p2p_detach()
    for es in ranks + [ref]:
        es.close()
    return bool(ok)


def envstep_case(rng):
    """One random population of step-wise envs (ses_env_reset / ses_env_step_generic) against the oracle's env objects."""
    which = rng.choice(["lander", "lander", "spread", "cartpole", "walker"])
    n = int(rng.choice([1, 7, 64, 90]))
    if which == "lander":
        pomdp, T = bool(rng.randint(0, 2)), int(rng.choice([20, 120]))
        es = HipES("LunarLanderContinuous-v2", 8, 4, False, False, pomdp=pomdp, max_step=300, eval_ep_num=1)
        init = rng.uniform(0, 1, (n, 16)).astype(np.float32)
        state, obs = es.env_reset(dev(init))
        sims = [co.LanderSim() for _ in range(n)]
        mask = np.array([0, 0, 1, 1, 0, 1, 0, 0], bool) if pomdp else np.zeros(8, bool)
        want = np.stack([s.reset(u) for s, u in zip(sims, init)])
        want[:, mask] = 0
        ok = np.array_equal(bits(obs.cpu().numpy()), bits(want))
        alive = np.ones(n, bool)
        for t in range(T):
            act = np.tanh(rng.randn(n, 4) * 1.5).astype(np.float32)
            o, r, d = (x.cpu().numpy() for x in es.env_step_generic(state, dev(act)))
            for i in np.flatnonzero(alive):
                wo, wr, wd = sims[i].step(float(act[i, 0]), float(act[i, 1]))
                wo[mask] = 0
                ok = ok and np.array_equal(bits(o[i]), bits(wo)) and bits(r[i:i + 1])[0] == bits(np.float32([wr]))[0] and bool(d[i]) == wd
                alive[i] = not wd
    elif which == "walker":
        n, T = min(n, 7), 30
        es = HipES("BipedalWalker-v3", 24, 4, False, False, max_step=300, eval_ep_num=1)
        init = rng.uniform(0, 1, (n, 4)).astype(np.float32)
        state, obs = es.env_reset(dev(init))
        sims = [co.WalkerSim() for _ in range(n)]
        want = np.stack([s.reset(u) for s, u in zip(sims, init)])
        ok = np.array_equal(bits(obs.cpu().numpy()), bits(want))
        alive = np.ones(n, bool)
        for t in range(T):
            act = np.tanh(rng.randn(n, 4)).astype(np.float32)
            o, r, d = (x.cpu().numpy() for x in es.env_step_generic(state, dev(act)))
            for i in np.flatnonzero(alive):
                wo, wr, wd = sims[i].step(act[i])
                ok = ok and np.array_equal(bits(o[i]), bits(wo)) and bits(r[i:i + 1])[0] == bits(np.float32([wr]))[0] and bool(d[i]) == wd
                alive[i] = not wd
    elif which == "spread":
        na = int(rng.choice([2, 3]))
        es = HipES("simple_spread", 6 * na, 5, True, False, max_step=25, eval_ep_num=1, n_agents=na)
        init = rng.uniform(-1, 1, (n, 4 * na)).astype(np.float32)
        state, obs = es.env_reset(dev(init))
        st = np.zeros((n, 6 * na), np.float32)
        st[:, : 2 * na], st[:, 4 * na:] = init[:, : 2 * na], init[:, 2 * na:]
        ok = True
        for t in range(25):
            act = rng.randint(0, 5, size=(n, na)).astype(np.int32)
            o, r, d = (x.cpu().numpy() for x in es.env_step_generic(state, dev(act)))
            for i in range(n):
                wr = co.spread_step(na, st[i], act[i])
                wo = np.concatenate([co.spread_obs(na, st[i], a) for a in range(na)])
                ok = ok and np.array_equal(bits(o[i]), bits(wo)) and bits(r[i:i + 1])[0] == bits(np.float32([wr]))[0] and bool(d[i]) == (t == 24)
    else:
        pomdp = bool(rng.randint(0, 2))
        es = HipES("CartPole-v1", 4, 2, True, False, pomdp=pomdp, max_step=500, eval_ep_num=1)
        init = rng.uniform(-0.05, 0.05, (n, 4)).astype(np.float32)
        state, obs = es.env_reset(dev(init))
        st = [init[:, k].copy() for k in range(4)]
        ret, status = np.zeros(n, np.float32), np.zeros(n, np.uint32)
        ok = True
        for t in range(80):
            act = rng.randint(0, 2, size=n).astype(np.int32)
            o, r, d = (x.cpu().numpy() for x in es.env_step_generic(state, dev(act)))
            co.cartpole_step_soa(1, 0, st[0], st[1], st[2], st[3], act, ret, status)
            want = np.stack(st, axis=1).astype(np.float32)
            if pomdp:
                want[:, [1, 3]] = 0
            term = (np.abs(st[0]) > 2.4) | (np.abs(st[2]) > 0.20943951)
            ok = ok and np.array_equal(bits(o), bits(want)) and np.array_equal(d.astype(bool), term)
    es.close()
    return bool(ok)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=200)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--time-limit", type=float, default=0.0, help="stop drawing cases after this many seconds (0 = never)")
    ap.add_argument("--only", default="", help="draw this kind only (e.g. sharded_tail)")
    ap.add_argument("--skip", default="", help="never draw this kind")
    args = ap.parse_args()
    import time
    t_start = time.time()
    t_progress = t_start
    rng = np.random.RandomState(args.seed)
    tally = {}
    done_cases = 0
    for case in range(args.cases):
        if args.time_limit and time.time() - t_start > args.time_limit:
            break
        done_cases += 1
        kind = rng.choice(["mlp", "mlp", "mlp", "gru", "gru", "gru_mfma", "lander", "lander", "lander_mlp", "lander_mlp", "walker",
                           "spread", "spread", "envstep", "sharded_tail"])
        if args.only:
            kind = args.only
        if kind == args.skip:
            done_cases -= 1
            continue
        mode = int(rng.randint(0, 2))
        shared = bool(rng.randint(0, 2))
        sigma = float(rng.choice([0.05, 0.3, 1.0, 3.0]))
        if kind == "mlp":
            n, E, T = int(rng.choice([1, 3, 17, 64, 129, 700, 1700, 2100])), int(rng.randint(1, 8)), int(rng.choice([1, 7, 60, 200]))
            pomdp, lpe, p64 = bool(rng.randint(0, 2)), int(rng.choice([0, 0, 1, 2, 4, 8, 16, 32])), bool(rng.rand() < 0.15)
            es = HipES("CartPole-v1", 4, 2, True, False, pomdp=pomdp, max_step=T, eval_ep_num=E, lanes_per_env=lpe, physics64=p64)
            es.set_tuning("rollout_packed", int(rng.choice([-1, 0, 1])))      # the packed step of lone waves: by rule, never, always
            es.set_tuning("rollout_mix_8_16", int(rng.rand() < 0.7))          # the (8, 16) mixed split (8193 ... ~12 000 envs)
            theta = (rng.randn(n, 226) * sigma).astype(np.float32)
            init = rng.uniform(-0.05, 0.05, (E, 4) if shared else (n, E, 4)).astype(np.float32)
            if rng.rand() < 0.2:
                init[..., 2] = rng.uniform(-1.5, 1.5, init[..., 2].shape)
            ref = co.rollout_cartpole(theta, init, E, T, obs_mask=0b1010 if pomdp else 0, physics64=p64)
            fit, ret, steps = es.rollout(dev(theta), dev(init), mode=mode, want_episodes=True)
            ok = np.array_equal(steps.cpu().numpy(), ref[2]) and np.array_equal(bits(fit.cpu().numpy()), bits(ref[0]))
            fit2 = es.rollout(dev(theta), dev(init), mode=mode)                 # the path without episode outputs
            ok = ok and np.array_equal(bits(fit2.cpu().numpy()), bits(ref[0]))
        elif kind in ("gru", "gru_mfma"):
            n, T = int(rng.choice([1, 5, 33, 130])), int(rng.choice([1, 9, 80]))
            E = int(rng.randint(12, 21)) if kind == "gru_mfma" else int(rng.randint(1, 12))
            pomdp = bool(rng.randint(0, 2))
            es = HipES("CartPole-v1", 4, 2, True, True, pomdp=pomdp, max_step=T, eval_ep_num=E)
            if rng.rand() < 0.5:                                        # force the lockstep / MFMA / sequential kernels at small sizes
                es.set_tuning("gru_ep_parallel_max", 0)
                es.set_tuning("gru_sequential", int(rng.rand() < 0.2))
                if rng.rand() < 0.3:
                    es.set_tuning("gru_mfma_min_e", 1)
                elif rng.rand() < 0.4:
                    es.set_tuning("gru_mfma4_min_e", 1)                  # the 4x4x1 MFMA step (taken for E <= 8)
            theta = (rng.randn(n, 6562) * min(sigma, 1.0)).astype(np.float32)
            init = rng.uniform(-0.05, 0.05, (E, 4) if shared else (n, E, 4)).astype(np.float32)
            ref = co.rollout_cartpole(theta, init, E, T, gru=True, obs_mask=0b1010 if pomdp else 0)
            fit, ret, steps = es.rollout(dev(theta), dev(init), mode=mode, want_episodes=True)
            ok = np.array_equal(steps.cpu().numpy(), ref[2]) and np.array_equal(bits(fit.cpu().numpy()), bits(ref[0]))
        elif kind in ("lander", "lander_mlp"):
            gru = kind == "lander"
            n, E, T = int(rng.choice([1, 6, 40])), int(rng.choice([1, 3, 5, 9, 13])), int(rng.choice([5, 60, 150]))
            pomdp = bool(rng.randint(0, 2))
            es = HipES("LunarLanderContinuous-v2", 8, 4, False, gru, pomdp=pomdp, max_step=T, eval_ep_num=E)
            es.set_tuning("box2d_lanes_per_env", int(rng.choice([0, 1, 2, 4, 8, 16, 32, 64])))     # MLP kernel only
            es.set_tuning("box2d_envs_per_wave", int(rng.choice([0, 0, 1, 3, 5, 10, 20, 33, 64])))  # (clamped to 64 / lanes per env)
            if gru and rng.rand() < 0.6:                                 # every GRU lander kernel: lockstep (1, 2, 4 offspring per wave), MFMA, sequential
                es.set_tuning("gru_ep_parallel_max", 0)
                es.set_tuning("lander_offspring_per_wave", int(rng.choice([0, 1, 2, 4])))
                es.set_tuning("gru_sequential", int(rng.rand() < 0.15))
                if rng.rand() < 0.2:
                    es.set_tuning("gru_mfma_min_e", 1)
            theta = (rng.randn(n, es.P) * min(sigma, 1.0)).astype(np.float32)
            if rng.rand() < 0.3:
                theta[:, -4:] += np.float32(1.5)                         # biased towards firing the main engine: long flights, soft touch-downs
            init = rng.uniform(0, 1, (E, 16) if shared else (n, E, 16)).astype(np.float32)
            ref = co.rollout_lander(theta, init, E, T, gru=gru, obs_mask=0b101100 if pomdp else 0)
            fit, ret, steps = es.rollout(dev(theta), dev(init), want_episodes=True)
            ok = (np.array_equal(steps.cpu().numpy(), ref[2]) and np.array_equal(bits(ret.cpu().numpy()), bits(ref[1])) and
                  np.array_equal(bits(fit.cpu().numpy()), bits(ref[0])))
        elif kind == "walker":
            n, E, T = int(rng.choice([1, 5, 17])), int(rng.choice([1, 2, 5])), int(rng.choice([5, 40, 90]))     # time-boxed: the oracle's walker step is ~1 ms
            es = HipES("BipedalWalker-v3", 24, 4, False, False, max_step=T, eval_ep_num=E)
            es.set_tuning("box2d_lanes_per_env", int(rng.choice([0, 1, 2, 4, 8, 16, 32, 64])))
            es.set_tuning("box2d_envs_per_wave", int(rng.choice([0, 0, 1, 3, 5, 10, 20, 33, 64])))
            theta = (rng.randn(n, es.P) * sigma).astype(np.float32)
            init = rng.uniform(0, 1, (E, 4) if shared else (n, E, 4)).astype(np.float32)
            ref = co.rollout_walker(theta, init, E, T)
            fit, ret, steps = es.rollout(dev(theta), dev(init), want_episodes=True)
            ok = (np.array_equal(steps.cpu().numpy(), ref[2]) and np.array_equal(bits(ret.cpu().numpy()), bits(ref[1])) and
                  np.array_equal(bits(fit.cpu().numpy()), bits(ref[0])))
        elif kind == "envstep":
            ok, n, E = envstep_case(rng), 0, 1
            es = None
        elif kind == "sharded_tail":
            ok, n, E = sharded_tail_case(rng), 0, 1
            es = None
        else:
            na = int(rng.choice([2, 3]))
            n, E = int(rng.choice([1, 9, 100, 515])), int(rng.randint(1, 7))
            es = HipES("simple_spread", 6 * na, 5, True, False, max_step=25, eval_ep_num=E, n_agents=na)
            theta = (rng.randn(n, es.P) * sigma).astype(np.float32)
            init = rng.uniform(-1, 1, (E, 4 * na) if shared else (n, E, 4 * na)).astype(np.float32)
            ref = co.rollout_spread(theta, init, E, na)
            fit, ret, _ = es.rollout(dev(theta), dev(init), want_episodes=True)
            ok = np.array_equal(bits(ret.cpu().numpy()), bits(ref[1])) and np.array_equal(bits(fit.cpu().numpy()), bits(ref[0]))
        if es is not None:
            es.close()
        t = tally.setdefault(kind, [0, 0])
        t[0] += 1
        t[1] += int(ok)
        if time.time() - t_progress > 60.0:                              # a sign of life for long passes (gpurun's silence guard)
            t_progress = time.time()
            print("progress", json.dumps({"cases": done_cases, "seconds": round(t_progress - t_start, 1),
                                          "all_bit_exact": all(v[0] == v[1] for v in tally.values())}), flush=True)
        if not ok:
            print("MISMATCH", json.dumps({"case": case, "kind": kind, "n": n, "E": E, "mode": mode, "shared": shared, "sigma": sigma}))
    print(json.dumps({"cases": done_cases, "seed": args.seed, "seconds": round(time.time() - t_start, 1), "by_kind": {k: {"cases": v[0], "bit_exact": v[1]} for k, v in tally.items()},
                      "all_bit_exact": all(v[0] == v[1] for v in tally.values())}))


if __name__ == "__main__":
    main()
