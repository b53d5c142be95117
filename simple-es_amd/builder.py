"""build_env / build_network / build_loop with the reference's signatures and YAML keys (builder.py:10-86).

Optional keys (all default to reference behaviour): strategy.noise ("philox" | "numpy"),
strategy.seed, env.seed, env.shared_init, env.n_agents (simple_spread: 2 like the reference, or 3),
env.physics ("float32" | "float64": gym-order float64 CartPole dynamics).
"""
from envs.gym_wrapper import GymWrapper
from envs.pettingzoo_wrapper import PettingzooWrapper
from learning_strategies.evolution.loop import ESLoop
from learning_strategies.evolution.offspring_strategies import openai_es, simple_evolution, simple_genetic
from networks.neural_network import GymEnvModel

_PETTINGZOO = ("simple_spread", "waterworld", "multiwalker")
_STRATEGIES = {
    "simple_evolution": (simple_evolution, ("init_sigma", "sigma_decay", "elite_num", "offspring_num")),
    "simple_genetic": (simple_genetic, ("init_sigma", "sigma_decay", "elite_num", "offspring_num")),
    "openai_es": (openai_es, ("init_sigma", "sigma_decay", "learning_rate", "offspring_num")),
}


def build_env(config):
    if config["name"] in _PETTINGZOO:
        return PettingzooWrapper(config["name"], config["max_step"], n_agents=config.get("n_agents", 2))
    return GymWrapper(config["name"], config["max_step"], config["pomdp"], physics=config.get("physics", "float32"))


def build_network(config):
    if config["name"] == "gym_model":
        return GymEnvModel(config["num_state"], config["num_action"], config["discrete_action"], config["gru"])
    raise ValueError(f"unknown network {config['name']!r}")


def build_strategy(strategy_cfg):
    if strategy_cfg["name"] not in _STRATEGIES:
        raise ValueError(f"unknown strategy {strategy_cfg['name']!r}")
    cls, keys = _STRATEGIES[strategy_cfg["name"]]
    extra = {k: strategy_cfg[k] for k in ("noise", "seed") if k in strategy_cfg}
    return cls(*[strategy_cfg[k] for k in keys], **extra)


def build_loop(config, gen_num, process_num, eval_ep_num, log, save_model_period):
    env = build_env(config["env"])
    network = build_network(config["network"])
    strategy = build_strategy(config["strategy"])
    return ESLoop(config, strategy, env, network, gen_num, process_num, eval_ep_num, log, save_model_period)
