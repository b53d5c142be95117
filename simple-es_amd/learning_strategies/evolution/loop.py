"""ESLoop -- generation loop with the reference's constructor and print/checkpoint behaviour
(learning_strategies/evolution/loop.py:14-125), with the per-generation hot path on the GPU:

    reference  p = mp.Pool(process_num); results = p.map(RolloutWorker, [(env, offspring, E), ...])
    here       one fused rollout kernel over this rank's shard of the population, then (multi-GPU) one
               all-gather of the fitness vector; `results` stays index-ordered like Pool.map.

`process_num` is accepted for command-line compatibility; the device path has no worker processes.
"""
import json
import os
import time
from collections import deque
from datetime import datetime

import torch
import torch.distributed as dist

from ses import HipES, MODE_EPISODIC, MODE_FIXED_LENGTH
from ses.parallel import all_ranks, attach_comm, comm_failed, comm_keep_going, comm_recover, world_size

from .abstracts import BaseESLoop


class _GenerationBatch:
    """The state of a run in the form ses_run_generations advances (include/ses.h: ses_gen_state): device buffers for the
    ping-pong halves, the strategy's constant parent map, and the host scalars.  `run(k)` enqueues k generations in one
    C call; `sync_back()` hands the result to the strategy object (so that checkpoints, get_elite_model() and a later
    per-generation call see exactly what they would have seen)."""

    K_MAX = 32

    @staticmethod
    def eligible(loop, strategy, population):
        from learning_strategies.evolution.offspring_strategies import openai_es, simple_evolution, simple_genetic
        hooked = any(name in loop.__dict__ or getattr(type(loop), name) is not getattr(ESLoop, name)
                     for name in ("rollout", "generation", "_init_states"))        # a caller observing the per-generation methods
        shard = population.shard
        if shard.world != world_size():
            return False
        if shard.world > 1:
            # sharded run: the C loop all-gathers the fitness itself, so a LIBRARY transport (peer stores or RCCL) has to
            # carry the shards; with torch.distributed as the only route the per-generation path stays
            owner = getattr(loop.dev, "_comm_owner", None)
            if owner is None:
                return False
            p2p_world, cap, rccl_world = owner.comm_route()
            if not ((p2p_world == shard.world and shard.per_rank <= cap) or rccl_world == shard.world):
                return False
        return (type(strategy) in (openai_es, simple_evolution, simple_genetic)
                and strategy.noise == "philox" and getattr(strategy, "fused", True) and hasattr(loop.dev, "run_generations")
                and not hooked and os.environ.get("SES_BATCH_GENERATIONS", "1") != "0")

    def __init__(self, loop, strategy, population):
        import numpy as np
        from learning_strategies.evolution.offspring_strategies import openai_es, simple_evolution
        from ses import _lib
        dev, P = loop.dev, strategy.P
        self.loop, self.strategy, self.dev = loop, strategy, dev
        shard = population.shard
        n = shard.n_global                  # population rows; this rank's theta holds shard.n_local of them
        n_loc = shard.n_local
        st = _lib.SesGenState()
        self.kind = (_lib.STRATEGY_OPENAI_ES if isinstance(strategy, openai_es) else
                     _lib.STRATEGY_SIMPLE_EVOLUTION if isinstance(strategy, simple_evolution) else _lib.STRATEGY_SIMPLE_GENETIC)
        st.strategy, st.n, st.mode = self.kind, n, loop.mode
        st.elite_num = 0 if self.kind == _lib.STRATEGY_OPENAI_ES else strategy.elite_num
        st.shared_init, st.init_width = int(loop.shared_init), dev.init_dim
        st.init_lo, st.init_hi = dev.init_range
        st.seed, st.env_seed = strategy.seed, loop.seed_env
        st.learning_rate = getattr(strategy, "learning_rate", 0.0)
        st.sigma_decay = strategy.sigma_decay
        st.sigma, st.pop_sigma = strategy.curr_sigma, strategy._last["sigma"]
        st.pop_gen = population.gen
        keep = self.keep = {}
        keep["theta"] = [population.theta.contiguous() if n_loc else dev.empty(1, P), dev.empty(max(n_loc, 1), P)]
        if self.kind == _lib.STRATEGY_OPENAI_ES:
            opt = strategy.optimizer
            st.adam_t = opt.t
            keep["parents"] = [strategy.mu_model.clone(), dev.empty(P)]
            keep["m"] = [opt.m.clone(), dev.empty(P)]
            keep["v"] = [opt.v.clone(), dev.empty(P)]
            self.map_host = strategy._last["idx_host"]
        elif self.kind == _lib.STRATEGY_SIMPLE_EVOLUTION:
            keep["parents"] = [strategy.mu_model.clone(), dev.empty(P)]       # elite[0] IS mu after every evaluate (and at the start)
            N = strategy.offspring_num

            def build():
                idx = np.zeros(N + 1, dtype=np.int32)
                idx[0], idx[1] = -1, -1
                return idx
            self.map_host = strategy._const_map(("evolution", N, True), build)
            keep["alias"] = strategy._alias_state
        else:
            keep["parents"] = [strategy.elite_models.clone().contiguous(), dev.empty(strategy.elite_num, P)]
            self.map_host = strategy._last["idx_host"]
        if self.kind != _lib.STRATEGY_OPENAI_ES:
            keep["map"] = torch.from_numpy(np.ascontiguousarray(self.map_host, dtype=np.int32)).to(dev.device)
            keep["wi"] = dev.empty(n + 3 * st.elite_num, dtype=torch.int32)
            keep["wf"] = dev.empty(st.elite_num, P)
            st.parent_map, st.work_i32, st.work_f32 = keep["map"].data_ptr(), keep["wi"].data_ptr(), keep["wf"].data_ptr()
            if "alias" in keep:
                st.alias_state = keep["alias"].data_ptr()
        keep["fitness"] = dev.empty(shard.per_rank * shard.world if shard.world > 1 else n)
        keep["init"] = dev.empty(1 if loop.shared_init else max(n_loc, 1), dev.E, dev.init_dim)
        st.fitness, st.init = keep["fitness"].data_ptr(), keep["init"].data_ptr()
        if shard.world > 1:
            # this rank's slot of the all-gather: the rollout writes the first n_local entries, a ragged tail stays -inf
            keep["fit_local"] = torch.full((shard.per_rank,), float("-inf"), dtype=torch.float32, device=dev.device)
            keep["comm"] = dev._comm_owner
            st.world, st.per_rank, st.n_local = shard.world, shard.per_rank, n_loc
            st.first_row = shard.rank * shard.per_rank
            st.comm, st.fit_local = keep["comm"]._h.value, keep["fit_local"].data_ptr()
        for i in (0, 1):
            st.theta[i], st.parents[i] = keep["theta"][i].data_ptr(), keep["parents"][i].data_ptr()
            if "m" in keep:
                st.adam_m[i], st.adam_v[i] = keep["m"][i].data_ptr(), keep["v"][i].data_ptr()
        st.cur = 0
        self.st = st
        self.shard = shard
        # two chunks in flight: pinned rings the kernels store the best reward / the time stamps into
        self.best = [torch.full((self.K_MAX,), float("nan"), dtype=torch.float32).pin_memory() for _ in range(2)]
        self.stamps = [torch.zeros(self.K_MAX, 2, dtype=torch.int64).pin_memory() for _ in range(2)]
        self.slot = 0
        # checkpoints without draining the queue: pinned copies of the elite vector, reused round-robin (at most two are waiting)
        self.ckpt_ring = [torch.empty(P, dtype=torch.float32).pin_memory() for _ in range(4)]
        self.ckpt_k = 0            # (a guarded run keeps up to comm_check_period / save_model_period of them waiting: ring grown on demand)

    def snapshot_elite(self, ep_num):
        """Enqueue, behind the generations issued so far, a copy of the elite's parameters (what get_elite_model() would
        return after sync_back()) into pinned host memory + an event: (generation, host vector, event).  The caller writes the
        checkpoint once the event has passed -- while the device is already working on the next chunk."""
        vec = self.keep["parents"][self.st.cur].view(-1, self.strategy.P)[0]
        need = self.loop.comm_check_period // max(self.loop.save_model_period, 1) + 4
        while len(self.ckpt_ring) < need:
            self.ckpt_ring.append(torch.empty(self.strategy.P, dtype=torch.float32).pin_memory())
        host = self.ckpt_ring[self.ckpt_k]
        self.ckpt_k = (self.ckpt_k + 1) % len(self.ckpt_ring)
        host.copy_(vec, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        return ep_num, host, ev

    def run(self, k):
        """Enqueue k generations; returns (pinned best[k], pinned stamps[k, 2], [curr_sigma after each generation])."""
        slot, self.slot = self.slot, self.slot ^ 1
        best, stamps = self.best[slot], self.stamps[slot]
        best[:k] = float("nan")
        stamps[:k].zero_()
        sigma, sigmas = self.st.sigma, []
        for _ in range(k):                              # what the C loop does to curr_sigma, for the per-generation prints
            sigma *= self.st.sigma_decay
            sigmas.append(sigma)
        self.dev.run_generations(self.st, k, best, stamps)
        return best, stamps, sigmas

    def sync_back(self):
        from learning_strategies.evolution.offspring_strategies import Population
        from ses import _lib
        st, s, keep = self.st, self.strategy, self.keep
        cur = st.cur
        s.curr_sigma = st.sigma
        s.gen = int(st.pop_gen) + 1
        parents = keep["parents"][cur]
        if self.kind == _lib.STRATEGY_OPENAI_ES:
            opt = s.optimizer
            s.mu_model, opt.m, opt.v, opt.t = parents, keep["m"][cur], keep["v"][cur], int(st.adam_t)
            opt.pi = s.mu_model
            s._spare = (keep["parents"][cur ^ 1], keep["m"][cur ^ 1], keep["v"][cur ^ 1])
        elif self.kind == _lib.STRATEGY_SIMPLE_EVOLUTION:
            s.mu_model = s.elite0 = parents
        else:
            s.elite_models = parents
        s._last = {"parents": parents.view(-1, s.P), "idx_host": self.map_host, "sigma": st.pop_sigma, "gen": int(st.pop_gen),
                   "shard": self.shard}
        return Population(keep["theta"][cur][: self.shard.n_local], self.shard, s.network, s.agent_ids, int(st.pop_gen))


class ESLoop(BaseESLoop):
    def __init__(self, config, offspring_strategy, env, network, generation_num, process_num, eval_ep_num,
                 log=False, save_model_period=10):
        super().__init__()
        self.env = env
        self.network = network
        self.process_num = process_num
        self.network.zero_init()
        self.offspring_strategy = offspring_strategy
        self.generation_num = generation_num
        self.eval_ep_num = eval_ep_num
        self.ep5_rewards = deque(maxlen=5)
        self.log = log
        self.save_model_period = save_model_period
        self.seed_env = int((config or {}).get("env", {}).get("seed", 0)) if isinstance(config, dict) else 0
        self.shared_init = bool((config or {}).get("env", {}).get("shared_init", False)) if isinstance(config, dict) else False
        env_cfg = (config or {}).get("env", {}) if isinstance(config, dict) else {}
        # env.fixed_length: the synchronous-benchmark mode of the rollout kernels (termination masked: finished envs
        # keep stepping, rewards gated; identical returns, data-independent work).  Default: episodic, like the reference.
        self.mode = MODE_FIXED_LENGTH if env_cfg.get("fixed_length", False) else MODE_EPISODIC
        self.env_variant = getattr(env, "variant", None)     # e.g. "box2d-restated": third-party physics restated, unpinned
        self.history = []
        self._metrics = None
        self._stamps = None
        self._ev_k = 0
        self._init_chunk = None     # ((first row, rows, shared), first generation, resets[gens, rows, E, W]) drawn ahead

        # logs/<env>/<timestamp>[_k]: the reference's makedirs (loop.py:40-47) raises when two loops start in the same
        # second; here the second one gets a suffix instead of silently sharing the directory.  Only rank 0 writes.
        rank = dist.get_rank() if (world_size() > 1 and dist.is_available() and dist.is_initialized()) else 0
        self.save_dir = None
        if rank == 0:
            stamp = datetime.now().strftime("%Y%m%d%H%M%S")
            base, k = f"logs/{self.env.name}/{stamp}", 0
            while True:
                self.save_dir = base if k == 0 else f"{base}_{k}"
                try:
                    os.makedirs(self.save_dir + "/saved_models/", exist_ok=False)
                    break
                except FileExistsError:
                    k += 1

        if self.log:
            import wandb                          # optional dependency, only when --log is given
            wandb.init(project=self.env.name, config=config)

        self.dev = HipES(env.name, network.num_state, network.num_action, network.discrete_action, network.use_gru,
                         pomdp=env.pomdp, max_step=env.horizon, eval_ep_num=eval_ep_num,
                         n_agents=getattr(env, "n_agents", 1), physics64=getattr(env, "physics64", False))
        attach_comm(self.dev)      # multi-GPU: the fitness all-gather runs on this handle's RCCL communicator
        # [end of the rollout phase, start of the kernel that writes the next population] in ticks of the GPU's 100 MHz
        # real-time counter, one pinned pair per generation in flight
        self._stamps = [torch.zeros(2, dtype=torch.int64).pin_memory() for _ in range(4)]
        self._prev_tail = 0
        self._tail_stamped = False
        self._guarded = False
        self.batched_generations = 0       # generations of run() that went through ses_run_generations (tests read it)

    def _init_states(self, gen, shard):
        """The env resets of generation `gen` for this rank's rows.  They depend on (env seed, generation, row) only, so they
        are drawn a chunk of generations ahead in ONE launch (ses_init_states_uniform_gens) and handed out as views: no
        reset kernel between one generation's last kernel and the next rollout.  (Round 2 drew generation g + 1 on a side
        stream during rollout g; side stream and main stream share a hardware queue, and the side kernel + its event sat
        between two kernels of the tail: ~5 us per generation, as much as it saved.)"""
        key = (shard.first, shard.n_local, self.shared_init)
        c = self._init_chunk
        if c is None or c[0] != key or not c[1] <= gen < c[1] + c[2].shape[0]:
            rows = 1 if self.shared_init else max(shard.n_local, 1)
            per_gen = rows * self.dev.E * self.dev.init_dim * 4
            gens = max(1, min(64, (32 << 20) // per_gen))
            buf = self.dev.init_states_uniform_gens(self.seed_env, gen, gens, 0 if self.shared_init else shard.first, rows,
                                                    shared=self.shared_init)
            c = self._init_chunk = (key, gen, buf)
        init = c[2][gen - c[1]]
        if self.shared_init:                       # common random numbers: every offspring sees the same resets
            return init[0]
        return init[: shard.n_local] if shard.n_local != init.shape[0] else init

    # the rollout phase of one generation: Population -> float32[N] fitness (identical on every rank)
    def rollout(self, population):
        shard = population.shard
        init = self._init_states(population.gen, shard)
        local = self.dev.rollout(population.theta, init, mode=self.mode) if shard.n_local else self.dev.empty(0)
        return shard.allgather_fitness(local, dev=self.dev)

    def generation(self, offsprings):
        """Enqueue one whole generation -- rollout of this rank's shard, fitness all-gather, strategy.evaluate with
        the next population -- WITHOUT waiting for the GPU.  Returns (next offspring group, best reward as a
        PendingReward, sigma, (event before, event after the rollout phase)).  run() is a loop of this plus the
        reference's logging; bench.py times exactly this method."""
        # timing without events (an event costs the launch stream ~5 us, three per generation were 5 % of it): the last
        # kernel of the rollout phase and the kernel that writes the next population stamp the GPU's real-time counter
        # into a pinned slot (ses_set_stamp); _report reads the slots of a finished generation
        stamp = self._stamps[self._ev_k]
        self._ev_k = (self._ev_k + 1) % len(self._stamps)
        stamp.zero_()
        self.dev.set_stamp(stamp[0:1])
        strategy = self.offspring_strategy
        sdev = getattr(strategy, "dev", None)
        # the tail stamp is written by the launch that writes the next population: there is none on a rank that owns no
        # rows of it (more ranks than offspring), and _report must not wait for one
        self._tail_stamped = sdev is not None and hasattr(sdev, "set_stamp") and offsprings.shard.n_local > 0
        if sdev is not None and hasattr(sdev, "set_stamp"):
            sdev.set_stamp(stamp[1:2])
        results = self.rollout(offsprings)
        if hasattr(strategy, "evaluate_async"):
            offsprings, best, curr_sigma = strategy.evaluate_async(results)
        else:                                      # a user strategy with the reference's synchronous evaluate() only
            offsprings, value, curr_sigma = strategy.evaluate(results)
            best = _Ready(value)
        return offsprings, best, curr_sigma, stamp

    def _report(self, ep_num, best, curr_sigma, stamp, start_time, tail_stamped, rank0):
        """The reference's per-generation bookkeeping (loop.py:85-99) for a generation whose results are in."""
        if self._guarded and comm_failed(self.dev):
            best_reward = float("nan")             # this rank's exchange timed out: the generations since the last boundary will
        else:                                      # be replayed (run()), there is nothing worth waiting for
            best_reward = best.result()            # waits for THAT generation only; the next one is already queued
        now = time.time()
        consumed_time = now - max(start_time, self._last_report)    # generations overlap: time between completions
        self._last_report = now
        if tail_stamped and int(stamp[1]) == 0:    # written a few us after the best reward: wait, bounded by TIME
            deadline = time.perf_counter() + 2e-3
            while int(stamp[1]) == 0 and time.perf_counter() < deadline:
                pass
        t_roll, t_tail = int(stamp[0]), int(stamp[1])
        if t_roll and t_tail and self._prev_tail and t_roll > self._prev_tail:
            rollout_consumed_time = (t_roll - self._prev_tail) * 1e-8      # GPU time of the rollout phase (100 MHz ticks)
            eval_consumed_time = max((t_tail - t_roll) * 1e-8, 0.0)        # GPU time of strategy.evaluate up to the next population
        else:                                      # first generation of a run / a strategy without a device handle
            eval_consumed_time = max((t_tail - t_roll) * 1e-8, 0.0) if t_roll and t_tail else 0.0
            rollout_consumed_time = max(consumed_time - eval_consumed_time, 0.0)
        self._prev_tail = t_tail
        self.history.append((best_reward, curr_sigma))
        self.ep5_rewards.append(best_reward)
        if not rank0:
            return
        ep5 = sum(self.ep5_rewards) / len(self.ep5_rewards)
        if self._metrics is None:                  # wandb-free metrics: same quantities as loop.py:94-99
            self._metrics = open(self.save_dir + "/metrics.jsonl", "a", buffering=1 << 16)
            self._metrics_flushed = now
            if self.env_variant:
                self._metrics.write(json.dumps({"env_variant": self.env_variant}) + "\n")
        self._metrics.write(json.dumps({"episode": ep_num, "best_reward": best_reward, "curr_sigma": curr_sigma,
                                        "ep5_mean_reward": ep5, "time": consumed_time,
                                        "rollout_t": rollout_consumed_time, "eval_t": eval_consumed_time}) + "\n")
        if now - self._metrics_flushed > 1.0:      # a reader (or a crash) is never more than a second behind; the flush
            self._metrics.flush()                  # happens while the GPU is busy with the next generation
            self._metrics_flushed = now
        print(f"episode: {ep_num}, Best reward: {best_reward:.2f}, sigma: {curr_sigma:.3f}, "
              f"time: {consumed_time:.2f}, rollout_t: {rollout_consumed_time:.2f}, "
              f"eval_t: {eval_consumed_time:.2f}")
        if self.log:
            import wandb
            wandb.log({"ep5_mean_reward": ep5, "curr_sigma": curr_sigma})

    def run(self):
        """The reference's generation loop (loop.py:52-104).  The host stays one generation ahead of the GPU: the
        prints / metrics of generation g are produced while generation g + 1 is running, so the device never waits
        for Python.  Every printed value is the one the reference would print for that generation.

        Multi-GPU: the peer-store all-gather marks an exchange whose peer did not answer within its time-out instead of
        waiting for ever.  At every boundary (a checkpoint generation, and at least every `comm_check_period`
        generations) the ranks agree whether that happened anywhere since the last boundary; if it did, every rank drops
        the transport (the all-gather continues on RCCL / torch.distributed), restores the strategy state of the last
        boundary and replays from there -- counter-based noise makes the replay bit-identical to an undisturbed run, and
        no checkpoint is written from a generation that consumed a NaN shard."""
        try:
            return self._run()
        finally:
            if self._metrics is not None:          # whatever ends the loop (an exception included) leaves the metrics on disk
                self._metrics.flush()

    comm_check_period = 64

    def _run(self):
        strategy = self.offspring_strategy
        offsprings = strategy.init_offspring(self.network, self.env.get_agent_ids())
        rank0 = offsprings.shard.rank == 0
        if rank0 and self.env_variant:
            print(f"note: {self.env.name} runs on this build's restatement of its third-party physics "
                  f"({self.env_variant}; parity with gym / Box2D is unpinned, see README)")
        guarded = offsprings.shard.world > 1 and hasattr(strategy, "snapshot") and getattr(strategy, "noise", "") == "philox"
        self._guarded = guarded
        # a guarded run owns a recovery (below): only then may a timed-out exchange be followed by further ones
        comm_keep_going(self.dev, guarded)
        try:
            return self._run_segments(strategy, offsprings, rank0, guarded)
        finally:
            comm_keep_going(self.dev, False)

    def _generation_batch(self, strategy, offsprings):
        """The device-side loop for this run, or None.  COLLECTIVE on a sharded run: eligible() looks at process-local state
        (a hooked method on one rank only, SES_BATCH_GENERATIONS in one rank's environment, this rank's transports), and the
        two forms issue DIFFERENT exchanges on the peer-store transport -- the C loop fuses the fitness exchange into a
        granule exchange, the per-generation path uses the flag-based all-gather plus one granule exchange -- so ranks that
        decided differently would wait for each other's exchanges until the time-out.  The ranks therefore take the MIN of
        their answers over the control plane: one rank that cannot, nobody does."""
        ok = _GenerationBatch.eligible(self, strategy, offsprings)
        shard = offsprings.shard
        if shard.world > 1:
            ok = all_ranks(ok, self.dev.device, shard.group)
        return _GenerationBatch(self, strategy, offsprings) if ok else None

    def _run_segments(self, strategy, offsprings, rank0, guarded):
        """The run as segments between boundaries -- checkpoint generations, the end, and for a guarded (multi-GPU) run at
        least every comm_check_period generations.  Inside a segment the generations go to the device k at a time through
        ses_run_generations (one C call instead of ~10 Python-level calls per generation; on several GPUs too: the fitness
        all-gather is issued by the C loop) when _GenerationBatch.eligible says so, one ESLoop.generation() at a time
        otherwise; both forms are bit-identical (tests/test_gpu_host_mirror.py, tests/test_gpu_multirank.py).  At a boundary
        every generation has been reported; a guarded run then agrees across the ranks whether an exchange timed out since
        the last boundary and, if so, rolls back to it."""
        period = self.save_model_period
        snap = (0, strategy.snapshot(offsprings), len(self.history), list(self.ep5_rewards)) if guarded else None
        batch = self._generation_batch(strategy, offsprings)
        self._last_report = 0.0
        # The device-side loop does not drain its queue at a checkpoint generation: the elite's parameters are copied to pinned
        # host memory by a copy enqueued behind that generation (snapshot_elite) and the file is written later -- an unguarded run
        # writes it when the chunk is reported (one chunk late, like the prints, while the device works on the next chunk); a
        # guarded run when the ranks have agreed, at the next boundary, that no exchange failed up to there (no checkpoint is
        # ever written from a generation that consumed a NaN shard), so its files appear up to comm_check_period generations
        # late.  (Draining cost conf/cartpole_openai.yaml, a checkpoint every 10 generations, 40 of its 280 us per generation.)
        ckpts = []
        drain = os.environ.get("SES_DRAIN_CHECKPOINTS", "0") == "1"       # (A/B runs of the old behaviour: drain at every checkpoint)

        def next_checkpoint(ep):
            return (ep // period + 1) * period

        self._reported_upto = 0
        try:
            return self._segments_loop(strategy, offsprings, rank0, guarded, period, snap, batch, ckpts, drain, next_checkpoint)
        finally:
            # Whatever ends the loop early (a failed launch, KeyboardInterrupt): the elite snapshots of generations that were
            # already REPORTED are good -- the reference would have written them synchronously (loop.py:101-104) -- and only wait
            # for their file.  (A guarded run writes at its agreement points; what is still waiting there is not known good.)
            if not guarded and ckpts and rank0:
                try:
                    self._write_checkpoints(ckpts, self._reported_upto, strategy)
                except Exception:
                    pass

    def _segments_loop(self, strategy, offsprings, rank0, guarded, period, snap, batch, ckpts, drain, next_checkpoint):
        chunk = None
        ep_num = 0
        while ep_num < self.generation_num:
            # where the queue is drained: the end; a guarded run's agreement points; per-generation segments and
            # SES_DRAIN_CHECKPOINTS=1 also at every checkpoint generation (their checkpoints are written synchronously)
            hard = self.generation_num
            if guarded:
                hard = min(hard, snap[0] + self.comm_check_period)
            if batch is None or drain:
                hard = min(hard, next_checkpoint(ep_num))
            if batch is not None:
                while ep_num < hard:
                    k = min(batch.K_MAX, hard - ep_num, next_checkpoint(ep_num) - ep_num)
                    t0 = time.time()
                    best, stamps, sigmas = batch.run(k)
                    self.batched_generations += k
                    ep_num += k
                    if not drain and rank0 and ep_num % period == 0:
                        ckpts.append(batch.snapshot_elite(ep_num))
                    if chunk is not None:
                        self._report_chunk(chunk, rank0)
                        if not guarded:
                            self._write_checkpoints(ckpts, chunk[0] + chunk[1] - 1, strategy)
                    chunk = (ep_num - k + 1, k, best, stamps, sigmas, t0)
                self._report_chunk(chunk, rank0)       # a boundary generation is reported before anything is written
                chunk = None
                if not guarded:
                    self._write_checkpoints(ckpts, ep_num, strategy)
                offsprings = batch.sync_back()
            else:
                pending = None
                while ep_num < hard:
                    ep_num += 1
                    start_time = time.time()
                    offsprings, best, curr_sigma, events = self.generation(offsprings)
                    if pending is not None:
                        self._report(*pending, rank0)
                    pending = (ep_num, best, curr_sigma, events, start_time, self._tail_stamped)
                self._report(*pending, rank0)
            if guarded:
                if comm_recover(self.dev):     # collective: some rank's exchange timed out since the last boundary
                    ep_num = snap[0]
                    offsprings = strategy.restore(snap[1])
                    del self.history[snap[2]:]
                    self.ep5_rewards.clear()
                    self.ep5_rewards.extend(snap[3])
                    self._prev_tail = 0
                    ckpts[:] = [c for c in ckpts if c[0] <= ep_num]      # snapshots of the generations that will be replayed
                    if rank0 and self._metrics is not None:      # the rows of the replayed generations above this one are void
                        self._metrics.write(json.dumps({"rollback_to": ep_num}) + "\n")
                    # the transport has changed: the device-side loop continues only if a library transport is left
                    batch = self._generation_batch(strategy, offsprings)
                    continue
                self._write_checkpoints(ckpts, ep_num, strategy)         # every rank's exchanges up to here were good
                snap = (ep_num, strategy.snapshot(offsprings), len(self.history), list(self.ep5_rewards))
            if ep_num % period == 0 and rank0 and (drain or batch is None):
                elite = strategy.get_elite_model()
                torch.save(elite.state_dict(), self.save_dir + "/saved_models" + f"/ep_{ep_num}.pt")
                self._metrics.flush()
        return offsprings

    def _write_checkpoints(self, ckpts, upto, strategy):
        """Write the checkpoints of the generations <= upto whose elite was snapshotted (loop.py:101-104)."""
        while ckpts and ckpts[0][0] <= upto:
            ep, host, ev = ckpts.pop(0)
            ev.synchronize()                       # the copy is behind generation `ep`, which has been reported: long done
            elite = strategy._model_from(host)
            torch.save(elite.state_dict(), self.save_dir + "/saved_models" + f"/ep_{ep}.pt")
            self._metrics.flush()

    def _report_chunk(self, chunk, rank0):
        from learning_strategies.evolution.offspring_strategies import PendingReward
        first, k, best, stamps, sigmas, t0 = chunk
        for j in range(k):
            self._report(first + j, PendingReward(best[j:j + 1]), sigmas[j], stamps[j], t0, True, rank0)
        self._reported_upto = first + k - 1

    def generations(self, offsprings, k):
        """Enqueue k generations without waiting for the GPU and without the per-generation bookkeeping of run(): through
        ses_run_generations when the run is eligible for it (bench.py's timed call on several GPUs), k x generation()
        otherwise.  Returns the next offspring group."""
        strategy = self.offspring_strategy
        owner = getattr(self.dev, "_comm_owner", None)
        route = owner.comm_route() if owner is not None else None       # (cached on the handle: no ctypes call)
        batch = getattr(self, "_bench_batch", None)
        if batch is None or batch[0] is not offsprings or batch[2] != route:
            # (collective on a sharded run: an all-reduce + .item(), i.e. a device sync with the nccl backend -- so the answer
            #  is kept, the negative one too, for as long as the caller hands back the population this method returned and the
            #  transports are the ones the answer was given for)
            batch = (offsprings, self._generation_batch(strategy, offsprings), route)
        if batch[1] is None:
            for _ in range(k):
                offsprings, _best, _sigma, _stamp = self.generation(offsprings)
            self._bench_batch = (offsprings, None, route)
            return offsprings
        done = 0
        while done < k:
            step = min(batch[1].K_MAX, k - done)
            batch[1].run(step)
            done += step
        pop = batch[1].sync_back()
        self._bench_batch = (pop, batch[1], route)
        return pop

    @property
    def device_side_loop(self):
        """True when the last generations() call went through ses_run_generations (bench.py reports it)."""
        b = getattr(self, "_bench_batch", None)
        return bool(b is not None and b[1] is not None)


class _Ready:
    def __init__(self, value):
        self._value = float(value)

    def result(self):
        return self._value


def RolloutWorker(arguments):
    """(env, {agent_id: model}, eval_ep_num) -> mean return, like the reference's worker (loop.py:108-125),
    evaluated by the fused rollout kernel on a population of one."""
    env, offspring, eval_ep_num = arguments
    model = next(iter(offspring.values()))
    dev = HipES(env.name, model.num_state, model.num_action, model.discrete_action, model.use_gru, pomdp=env.pomdp,
                max_step=env.horizon, eval_ep_num=eval_ep_num, n_agents=getattr(env, "n_agents", 1),
                physics64=getattr(env, "physics64", False))
    theta = torch.from_numpy(model.flat()[None, :]).to(dev.device)
    init = dev.init_states_uniform(getattr(env, "seed_env", 0), getattr(env, "_episode", 0), 0, 1)
    env._episode = getattr(env, "_episode", 0) + 1
    fit = dev.rollout(theta, init)
    out = float(fit[0].item())
    dev.close()
    return out
