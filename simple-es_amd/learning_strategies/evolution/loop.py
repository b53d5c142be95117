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

from ses import HipES, MODE_EPISODIC

from .abstracts import BaseESLoop


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
        self.history = []
        self._metrics = None

        stamp = datetime.now().strftime("%Y%m%d%H%M%S")
        self.save_dir = f"logs/{self.env.name}/{stamp}"
        os.makedirs(self.save_dir + "/saved_models/", exist_ok=True)

        if self.log:
            import wandb                          # optional dependency, only when --log is given
            wandb.init(project=self.env.name, config=config)

        self.dev = HipES(env.name, network.num_state, network.num_action, network.discrete_action, network.use_gru,
                         pomdp=env.pomdp, max_step=env.horizon, eval_ep_num=eval_ep_num,
                         n_agents=getattr(env, "n_agents", 1), physics64=getattr(env, "physics64", False))

    # the rollout phase of one generation: Population -> float32[N] fitness (identical on every rank)
    def rollout(self, population):
        shard = population.shard
        if self.shared_init:                       # common random numbers: every offspring sees the same resets
            init = self.dev.init_states_uniform(self.seed_env, population.gen, 0, 1, shared=True)[0].contiguous()
        else:                                      # reference behaviour: independent resets per offspring
            init = self.dev.init_states_uniform(self.seed_env, population.gen, shard.first, max(shard.n_local, 1))
            init = init[: shard.n_local].contiguous()
        local = self.dev.rollout(population.theta, init, mode=MODE_EPISODIC) if shard.n_local else self.dev.empty(0)
        return shard.allgather_fitness(local)

    def run(self):
        offsprings = self.offspring_strategy.init_offspring(self.network, self.env.get_agent_ids())
        rank0 = offsprings.shard.rank == 0
        ep_num = 0
        for _ in range(self.generation_num):
            start_time = time.time()
            ep_num += 1

            # rollout_t is the GPU time of the rollout phase (HIP events, no host wait); evaluate() ends with the one
            # device read-back of a generation (best reward), so eval_t = everything else in the wall time
            ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            ev0.record()
            results = self.rollout(offsprings)
            ev1.record()
            offsprings, best_reward, curr_sigma = self.offspring_strategy.evaluate(results)
            ev1.synchronize()
            consumed_time = time.time() - start_time
            rollout_consumed_time = ev0.elapsed_time(ev1) * 1e-3
            eval_consumed_time = max(consumed_time - rollout_consumed_time, 0.0)
            self.history.append((best_reward, curr_sigma))
            self.ep5_rewards.append(best_reward)
            if rank0:                                   # wandb-free metrics: same quantities as loop.py:94-99
                if self._metrics is None:
                    self._metrics = open(self.save_dir + "/metrics.jsonl", "a", buffering=1)   # line-buffered
                self._metrics.write(json.dumps({"episode": ep_num, "best_reward": best_reward, "curr_sigma": curr_sigma,
                                                "ep5_mean_reward": sum(self.ep5_rewards) / len(self.ep5_rewards),
                                                "time": consumed_time, "rollout_t": rollout_consumed_time,
                                                "eval_t": eval_consumed_time}) + "\n")
            if rank0:
                print(f"episode: {ep_num}, Best reward: {best_reward:.2f}, sigma: {curr_sigma:.3f}, "
                      f"time: {consumed_time:.2f}, rollout_t: {rollout_consumed_time:.2f}, "
                      f"eval_t: {eval_consumed_time:.2f}")

            if self.log and rank0:
                import wandb
                wandb.log({"ep5_mean_reward": sum(self.ep5_rewards) / len(self.ep5_rewards),
                           "curr_sigma": curr_sigma})

            if ep_num % self.save_model_period == 0 and rank0:
                elite = self.offspring_strategy.get_elite_model()
                torch.save(elite.state_dict(), self.save_dir + "/saved_models" + f"/ep_{ep_num}.pt")


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
