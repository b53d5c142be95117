#!/usr/bin/env python3
"""Summarise the SQ counter passes of tools/prof_sq.sh / prof_sq_c3.sh (rocprofv3 --pmc, kernel-trace only) into
profiles/<tag>_sq_<name>.json: per-dispatch averages of every counter for the kernels whose name contains <match>, the
sha256 of their machine code (tools/kernel_hash.py) and a few derived figures.

    python tools/collect_sq.py r02 rollout k_rollout_cartpole_mlp gpurun_out/sq_1 gpurun_out/sq_2
"""
import collections
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import kernel_hash  # noqa: E402

PEAK_VALU = 1024 * 2.4e9 / 2      # wave64 VALU instructions per second: 1024 SIMDs x 2.4 GHz, 2 cycles each at best


def main():
    tag, name, match = sys.argv[1:4]
    dirs = sys.argv[4:]
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    dur = collections.defaultdict(list)
    for d in dirs:
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                k = r["Kernel_Name"].split("(")[0]
                if match in k:
                    agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                k = r["Kernel_Name"].split("(")[0]
                if match in k:
                    dur[k].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-3)
    out = {"command": "rocprofv3 --pmc <8 SQ counters> --kernel-trace (two passes; tools/prof_sq.sh or prof_sq_c3.sh), "
                      "summarised by tools/collect_sq.py", "kernels": {}}
    for k, d in agg.items():
        per = {c: sum(v) / len(v) for c, v in sorted(d.items())}
        rec = {"dispatches": max(len(v) for v in d.values()), "per_dispatch": per,
               "avg_us_under_counters": sum(dur[k]) / len(dur[k]) if dur[k] else None}
        if "SQ_INSTS_VALU" in per and "SQ_WAVES" in per:
            rec["valu_instructions_per_wave"] = per["SQ_INSTS_VALU"] / per["SQ_WAVES"]
        if "SQ_WAVE_CYCLES" in per and "SQ_INSTS_VALU" in per:
            rec["wave_cycles_per_valu_instruction"] = 4.0 * per["SQ_WAVE_CYCLES"] / per["SQ_INSTS_VALU"]   # counter in quad-cycles
        if "SQ_WAIT_INST_ANY" in per and "SQ_WAVE_CYCLES" in per:
            rec["wait_inst_any_frac_of_wave_cycles"] = per["SQ_WAIT_INST_ANY"] / per["SQ_WAVE_CYCLES"]
        if "SQ_ACTIVE_INST_ANY" in per and "SQ_WAVE_CYCLES" in per:
            rec["active_inst_any_frac_of_wave_cycles"] = per["SQ_ACTIVE_INST_ANY"] / per["SQ_WAVE_CYCLES"]
        out["kernels"][k] = rec
    out["peak_valu_wave_instr_per_s"] = PEAK_VALU
    out["kernel_match"] = match           # the name fragment the hash is taken over (tests/test_profiles_current.py)
    out["kernel_code_sha256"] = kernel_hash.hash_kernels(os.path.join(ROOT, "simple-es_amd", "libses_hip.so"), match)
    # bench.py reads per_dispatch of the first (dominant) kernel at the top level
    if out["kernels"]:
        first = max(out["kernels"].items(), key=lambda kv: kv[1]["per_dispatch"].get("SQ_INSTS_VALU", 0))
        out["kernel"] = first[0]
        out["per_dispatch"] = first[1]["per_dispatch"]
    path = os.path.join(ROOT, "profiles", f"{tag}_sq_{name}.json")
    json.dump(out, open(path, "w"), indent=1)
    print(path, json.dumps({k: {"valu/wave": v.get("valu_instructions_per_wave"), "us": v["avg_us_under_counters"]}
                            for k, v in out["kernels"].items()}))


if __name__ == "__main__":
    main()
