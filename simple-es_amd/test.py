"""Headless checkpoint playback: python test.py --cfg-path conf/cartpole.yaml --ckpt-path logs/.../ep_10.pt

Counterpart of the reference's test.py (test.py:20-68) without the renderer: loads a state_dict written by
either code base, plays 100 episodes and prints `reward: ... ep_step: ...` per episode.  The 100 episodes are ONE
launch of the fused rollout kernel (a population of one offspring with eval_ep_num = 100); --save-gif is not
available on the device path.
"""
import argparse

import torch
import yaml

import builder
from ses import HipES


def main():
    parser = argparse.ArgumentParser()
    parser.add_argument("--cfg-path", type=str, default="conf/cartpole.yaml")
    parser.add_argument("--ckpt-path", type=str, required=True)
    parser.add_argument("--episodes", type=int, default=100)
    parser.add_argument("--seed", type=int, default=None, help="env seed (default: env.seed of the config, else 0)")
    parser.add_argument("--save-gif", action="store_true")
    args = parser.parse_args()
    if args.save_gif:
        raise SystemExit("--save-gif needs a renderer; the device envs have none")

    with open(args.cfg_path) as f:
        config = yaml.load(f, Loader=yaml.FullLoader)
    env = builder.build_env(config["env"])
    seed = args.seed if args.seed is not None else int(config["env"].get("seed", 0))
    network = builder.build_network(config["network"])
    network.load_state_dict(torch.load(args.ckpt_path))

    dev = HipES(env.name, network.num_state, network.num_action, network.discrete_action, network.use_gru,
                pomdp=env.pomdp, max_step=env.horizon, eval_ep_num=args.episodes, n_agents=getattr(env, "n_agents", 1),
                physics64=getattr(env, "physics64", False))       # replay on the dynamics the policy was trained on
    theta = torch.from_numpy(network.flat()[None, :]).to(dev.device)
    init = dev.init_states_uniform(seed, 0, 0, 1)
    fit, ep_ret, ep_steps = dev.rollout(theta, init, want_episodes=True)
    rets = ep_ret[0].cpu().tolist()
    steps = ep_steps[0].cpu().tolist() if ep_steps is not None else [env.horizon] * args.episodes
    for r, s in zip(rets, steps):
        print("reward: ", r, "ep_step: ", s)
    print(f"mean reward over {args.episodes} episodes: {float(fit[0]):.3f}")


if __name__ == "__main__":
    main()
