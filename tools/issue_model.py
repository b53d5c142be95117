#!/usr/bin/env python3
"""Serial-issue model of the headline rollout kernel, written next to the profiles (no GPU needed).

    python tools/issue_model.py profiles/r05_issue_model.json

Compiles csrc/ses_rollout.hip to a gfx950 listing with the build's flags, prices the two loop bodies the headline job
runs (fully observed CartPole: the light wave at 16 lanes per env, the heavy wave at 4) with tools/loop_issue_cost.py --
every VALU instruction at the issue cadence tools/valu_issue.hip / vgpr_bank.hip measured on MI355X (2, 4 or 8 cycles) --
and records them with the hash of the kernel's machine code in the built library.  bench.py turns it into
`rollout_kernel.valu_issue_model_frac`: SIMD cycles the loops need if every instruction issued alone at its measured
cadence, over the SIMD cycles the kernel had.  It says how far the kernel is from the ceiling of its own instruction MIX;
`valu_issue_frac` beside it is the nominal one (every instruction at 2 cycles)."""
import json
import os
import re
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
import kernel_hash  # noqa: E402
import loop_issue_cost  # noqa: E402

KERNEL = "k_rollout_cartpole_mlp_mixILb1ELi16"       # <fixed_length = true, light lanes per env = 16>: what bench.py's headline launches


def build_flags():
    text = open(os.path.join(ROOT, "simple-es_amd", "csrc", "build.sh")).read()
    m = re.search(r"FLAGS=\((.*?)\)", text, re.S)
    return m.group(1).split()


def main():
    out_path = sys.argv[1]
    src = os.path.join(ROOT, "simple-es_amd", "csrc", "ses_rollout.hip")
    with tempfile.TemporaryDirectory() as tmp:
        lst = os.path.join(tmp, "rollout.s")
        subprocess.check_call(["/opt/rocm/bin/hipcc"] + build_flags() + ["--cuda-device-only", "-S", src, "-o", lst],
                              stderr=subprocess.DEVNULL)
        name, found = loop_issue_cost.loops(open(lst).read().splitlines(), KERNEL)
    heavy = [lp for lp in found if lp["valu"] >= 120]
    light = [lp for lp in found if lp["valu"] < 120]
    assert len(heavy) == 2 and len(light) == 2, [(lp["label"], lp["valu"]) for lp in found]
    # each body exists twice: with the observation mask applied (POMDP: four more v_cndmask_b32) and without -- the headline is fully observed
    pick = lambda pair: min(pair, key=lambda lp: lp["kinds"].get("v_cndmask_b32", 0))   # noqa: E731
    h, l = pick(heavy), pick(light)
    lib = os.path.join(ROOT, "simple-es_amd", "libses_hip.so")
    model = {
        "kernel": name,
        "kernel_match": "k_rollout_cartpole_mlp",
        "kernel_code_sha256": kernel_hash.hash_kernels(lib, "k_rollout_cartpole_mlp"),
        "workload": "4096 offspring x 5 episodes x 500 fixed-length steps: 1024 light waves (4 envs at 16 lanes per env) + 1024 heavy "
                    "waves (16 envs at 4 lanes per env), one of each per SIMD",
        "issue_cycles_per_step": {"lanes_per_env_16": {"waves": 1024, "cycles": l["cycles"], "valu": l["valu"]},
                                  "lanes_per_env_4": {"waves": 1024, "cycles": h["cycles"], "valu": h["valu"]}},
        "clock_ghz_under_load": 2.4,
        "clock_source": "DESIGN 6: 2394-2400 MHz sampled during the bench",
        "pricing": "tools/loop_issue_cost.py: 2 cycles for mul/add/sub/mov/and/lshr and fma forms with <= 2 register sources or three "
                   "registers of mixed parity, 8 for v_rcp_f32, 4 for everything else (profiles/r01_valu_issue.txt, r01_vgpr_bank.txt)",
    }
    json.dump(model, open(out_path, "w"), indent=1)
    print(json.dumps(model["issue_cycles_per_step"]))


if __name__ == "__main__":
    main()
