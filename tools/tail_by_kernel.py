#!/usr/bin/env python3
"""Per-kernel durations of the openai_es tail in a rocprofv3 kernel trace of tools/time_tail.py (SES_TAIL_SHAPES=8x4096,8x8192):
the trace holds, in order, the replicated and the sharded form of each shape; kernels are told apart by name and grid.
usage: tail_by_kernel.py <kernel_trace.csv>"""
import csv, collections, statistics, sys

rows = list(csv.DictReader(open(sys.argv[1])))
by = collections.defaultdict(list)
for r in rows:
    name = r["Kernel_Name"].split("(")[0].replace("ses::", "").replace("void ", "")
    grid = (int(r["Grid_Size_X"]) // max(int(r["Workgroup_Size_X"]), 1), int(r["Grid_Size_Y"]) // max(int(r["Workgroup_Size_Y"]), 1))
    by[(name, grid)].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-3)
print(f"{'kernel':58s} {'grid (workgroups)':>18s} {'launches':>8s} {'median us':>10s} {'mean us':>9s}")
for (name, grid), v in sorted(by.items(), key=lambda kv: (kv[0][0], kv[0][1])):
    if len(v) < 20:
        continue
    print(f"{name[:58]:58s} {str(grid):>18s} {len(v):8d} {statistics.median(v):10.2f} {statistics.mean(v):9.2f}")
