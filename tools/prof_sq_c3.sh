#!/bin/bash
# SQ counters of the C3 rollout (LunarLanderContinuous-v2 POMDP GRU, 4096 offspring x 5 episodes x <= 300 steps):
# two passes of <= 8 SQ counters over tools/time_c3.py, kernel-trace only.  Summary on stdout (copy to profiles/).
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY" \
           "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_INSTS_FLAT GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  rm -rf $R/gpurun_out/sqc3_$i
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $R/gpurun_out/sqc3_$i -- python3 $R/tools/time_c3.py "${1:-4096}" gru > $R/gpurun_out/sqc3_$i.log 2>&1
  f=$(find $R/gpurun_out/sqc3_$i -name "*counter_collection.csv" | head -1)
  python3 - "$f" <<'PY'
import csv, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"].split("(")[0]
    if "rollout" in k:
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in agg.items():
    print(k)
    for c, v in sorted(d.items()):
        print(f"   {c:26s} avg per dispatch = {sum(v)/len(v):18.1f}  (n={len(v)})")
PY
done
tail -2 $R/gpurun_out/sqc3_1.log
