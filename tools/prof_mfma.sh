#!/bin/bash
# SQ counters of the MFMA GRU rollout kernel (eval_ep_num 16): matrix-pipe busy cycles next to the VALU picture.
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_LDS" \
           "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INST_CYCLES_SALU SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE SQ_ACTIVE_INST_LDS"; do
  i=$((i+1))
  rm -rf $R/gpurun_out/mf_$i
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $R/gpurun_out/mf_$i -- python3 $R/bench.py --gru --eval-ep-num 16 --no-cpu-baseline --no-roofline --steps 3 --warmup 1 "$@" > $R/gpurun_out/mf_$i.log 2>&1
  f=$(find $R/gpurun_out/mf_$i -name "*counter_collection.csv" | head -1)
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
        print(f"   {c:28s} avg per dispatch = {sum(v)/len(v):18.1f}  (n={len(v)})")
PY
done
