#!/usr/bin/env python3
"""Harvest trained parameter vectors for fixture G9 (long-lived policies) -- runs on the GPU box.

Three product runs (the drop-in ESLoop over the HIP path), checkpoints every few generations, and the elite
vectors of those checkpoints stacked into ONE npz of plain float32 arrays:

    mlp     [k, 226]    CartPole-v1, MLP, simple_evolution (conf/cartpole.yaml) + openai_es (conf/cartpole_openai.yaml, 256 offspring)
    gru     [k, 6562]   POMDP CartPole-v1, GRU, simple_evolution (conf/cartpole_pomdp_gru.yaml -- README.md:42 of the reference)
    lander  [k, 6756]   LunarLanderContinuous-v2 POMDP, GRU, openai_es (conf/lunarlander_openai.yaml, larger population)
    *_gen               the generation each vector was saved at, *_best the best return of that generation

These are INPUTS only (where a theta comes from does not matter to the fixture); the returns of fixture G9 are
produced by the imported reference in tests/golden/make_golden.py g9.

    python tools/g9_train.py gpurun_out/g9_seeds.npz
    python tools/g9_train.py gpurun_out/g10_seeds.npz box2d      # lander_mlp [k, 420], walker_mlp [k, 932]: fixture G10
    python tools/g9_train.py gpurun_out/g7t_seeds.npz spread     # spread2 [k, 581], spread3 [k, 773]: fixture G7t
"""
import contextlib
import glob
import io
import os
import sys
import tempfile
import time

import numpy as np
import torch
import yaml

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "simple-es_amd")]
import builder  # noqa: E402

KEYS_MLP = ["fc1.weight", "fc1.bias", "fc2.weight", "fc2.bias"]
KEYS_GRU = ["fc1.weight", "fc1.bias", "gru.weight_ih_l0", "gru.weight_hh_l0", "gru.bias_ih_l0", "gru.bias_hh_l0",
            "fc2.weight", "fc2.bias"]


def train(cfg_name, gens, period, episodes=5, three_agents=False, **override):
    cfg = yaml.load(open(os.path.join(ROOT, "simple-es_amd", "conf", cfg_name)), Loader=yaml.FullLoader)
    for k, v in override.items():
        cfg["strategy"][k] = v
    if three_agents:                                 # simple_spread with three agents: 18 observations (BASELINE configs[4])
        cfg["env"]["n_agents"] = 3
        cfg["network"]["num_state"] = 18
        cfg["strategy"].pop("learning_rate", None)
    cfg["strategy"].setdefault("seed", 0)
    cfg["env"].setdefault("seed", 0)
    os.chdir(tempfile.mkdtemp())
    loop = builder.build_loop(cfg, gens, 1, episodes, False, period)
    t0 = time.time()
    with contextlib.redirect_stdout(io.StringIO()):
        loop.run()
    best = [b for b, _ in loop.history]
    keys = KEYS_GRU if cfg["network"]["gru"] else KEYS_MLP
    vecs, at = [], []
    for path in sorted(glob.glob(os.path.join(loop.save_dir, "saved_models", "ep_*.pt")),
                       key=lambda p: int(os.path.basename(p)[3:-3])):
        sd = torch.load(path, map_location="cpu")
        assert list(sd.keys()) == keys, list(sd.keys())
        vecs.append(np.concatenate([sd[k].numpy().reshape(-1) for k in keys]).astype(np.float32))
        at.append(int(os.path.basename(path)[3:-3]))
    print(f"{cfg_name} {override}: {gens} generations in {time.time() - t0:.1f}s; best per {max(gens // 20, 1)}:",
          [round(max(best[i:i + max(gens // 20, 1)]), 1) for i in range(0, gens, max(gens // 20, 1))], flush=True)
    return np.stack(vecs), np.array(at, dtype=np.int32), np.array([best[min(a, gens) - 1] for a in at], dtype=np.float32)


def main():
    out_path = os.path.abspath(sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/g9_seeds.npz")
    os.makedirs(os.path.dirname(out_path), exist_ok=True)
    out = {}

    def put(tag, parts):
        out[tag] = np.concatenate([p[0] for p in parts])
        out[tag + "_gen"] = np.concatenate([p[1] for p in parts])
        out[tag + "_best"] = np.concatenate([p[2] for p in parts])

    if len(sys.argv) > 2 and sys.argv[2] == "spread":
        # trained simple_spread teams for G7's extension: conf/simplespread.yaml (2 agents, openai_es) and the BASELINE shape
        # (3 agents, simple_evolution)
        put("spread2", [train("simplespread.yaml", 300, 15, episodes=5, offspring_num=512)])
        cfg3 = {"name": "simple_evolution", "init_sigma": 1.0, "sigma_decay": 0.999, "elite_num": 16, "offspring_num": 512}
        put("spread3", [train("simplespread.yaml", 300, 15, episodes=5, three_agents=True, **cfg3)])
        np.savez_compressed(out_path, **out)
        print("wrote", out_path, {k: v.shape for k, v in out.items()})
        return
    if len(sys.argv) > 2 and sys.argv[2] == "box2d":
        # fixture G10: the MLP policies of the reference's two Box2D configs (conf/lunarlander.yaml: 8-32-4, conf/bipedalwalker.yaml:
        # 24-32-4), elite checkpoints of product runs at the configs' own strategies, larger populations
        put("lander_mlp", [train("lunarlander.yaml", 240, 10, episodes=3, offspring_num=512)])
        put("walker_mlp", [train("bipedalwalker.yaml", 160, 8, episodes=3, offspring_num=480)])
        np.savez_compressed(out_path, **out)
        print("wrote", out_path, {k: v.shape for k, v in out.items()})
        return

    put("mlp", [train("cartpole.yaml", 60, 3), train("cartpole_openai.yaml", 60, 4, offspring_num=256)])
    put("gru", [train("cartpole_pomdp_gru.yaml", 160, 8)])
    lander_gens = int(os.environ.get("G9_LANDER_GENS", "600"))
    put("lander", [train("lunarlander_openai.yaml", lander_gens, max(lander_gens // 24, 1), episodes=3, offspring_num=1024)])
    np.savez_compressed(out_path, **out)
    print("wrote", out_path, {k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main()
