#!/usr/bin/env python3
"""Turn the two rocprofv3 --pmc passes of tools/prof_pmc.sh (gpurun_out/pmc_FETCH_SIZE, pmc_WRITE_SIZE) into
profiles/<round>_pmc_env_step.json (round tag = argv[1], default r02): HBM bytes per launch of the env-step kernel,
with the gfx950 FETCH_SIZE correction and the sha256 of the kernel's machine code (tools/kernel_hash.py) -- bench.py
attaches the traffic figure to its line only while that hash matches the library it runs."""
import collections
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import kernel_hash  # noqa: E402

TAG = sys.argv[1] if len(sys.argv) > 1 else "r03"
out = {"command": "rocprofv3 --pmc {FETCH_SIZE|WRITE_SIZE} --kernel-trace -- python3 bench.py --no-cpu-baseline --steps 5 "
                  "--warmup 1 (two separate passes, tools/prof_pmc.sh; summarised by tools/collect_pmc.py)",
       "kernel": "void ses::k_env_step_cartpole_v4<true>", "n_env": 1 << 24, "all_kernels": {}}
for counter in ("FETCH_SIZE", "WRITE_SIZE"):
    files = glob.glob(os.path.join(ROOT, "gpurun_out", "pmc_" + counter, "**", "*counter_collection.csv"), recursive=True)
    if not files:
        sys.exit("missing pass " + counter)
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(max(files, key=os.path.getmtime))):      # the most recent pass
        if r.get("Counter_Name") == counter:
            agg[r["Kernel_Name"].split("(")[0]].append(float(r["Counter_Value"]))
    out["all_kernels"][counter] = {k: {"dispatches": len(v), "avg_KB": sum(v) / len(v)} for k, v in agg.items()
                                   if k.startswith("ses::") or k.startswith("void ses::")}
    out[counter + "_avg_KB"] = out["all_kernels"][counter][out["kernel"]]["avg_KB"]
out["gfx950_correction"] = ("FETCH_SIZE counts 128-B requests at 64 B for 16 B/lane streaming reads: x2 "
                            "(MI355X_MICROARCH.md, HBM)")
out["traffic_bytes_per_launch"] = (2.0 * out["FETCH_SIZE_avg_KB"] + out["WRITE_SIZE_avg_KB"]) * 1024.0
out["algorithmic_bytes_per_launch"] = 52 * out["n_env"]
out["traffic_over_algorithmic"] = out["traffic_bytes_per_launch"] / out["algorithmic_bytes_per_launch"]
out["kernel_match"] = "k_env_step_cartpole_v4"
out["kernel_code_sha256"] = kernel_hash.hash_kernels(os.path.join(ROOT, "simple-es_amd", "libses_hip.so"),
                                                     "k_env_step_cartpole_v4")
json.dump(out, open(os.path.join(ROOT, "profiles", TAG + "_pmc_env_step.json"), "w"), indent=1)
print(json.dumps({k: out[k] for k in ("FETCH_SIZE_avg_KB", "WRITE_SIZE_avg_KB", "traffic_bytes_per_launch",
                                      "traffic_over_algorithmic")}))
