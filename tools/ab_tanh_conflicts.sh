#!/bin/bash
# Prices the bank conflicts of the tanh-table reads in the fused CartPole MLP rollout: the shipped library
# against ab/libNOCONF.so (csrc/build.sh -DSES_EXPERIMENT_TANH_NO_CONFLICT: every lane's entry index forced into its own 16-byte
# bank group -- WRONG results, same instruction stream + one v_bfi per read), interleaved, then one SQ counter pass each.
#   built beforehand: SES_OUT=$PWD/ab/libNOCONF.so SES_OBJ=/tmp/objNC bash simple-es_amd/csrc/build.sh -DSES_EXPERIMENT_TANH_NO_CONFLICT
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out
for rep in 1 2 3; do
  for lib in simple-es_amd/libses_hip.so ab/libNOCONF.so; do
    SES_LIB_PATH=$R/$lib python3 $R/bench.py --steps 300 --warmup 50 --blocks 9 --no-cpu-baseline --no-roofline --no-extras 2>/dev/null \
      | python3 -c "import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print('$lib', 'ms_per_generation', round(d['ms_per_step'],5), 'rollout_call_ms', round(d['rollout_kernel']['ms'],5))"
  done
done
for lib in simple-es_amd/libses_hip.so ab/libNOCONF.so; do
  rm -rf $R/gpurun_out/sq_tc
  SES_LIB_PATH=$R/$lib rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_INSTS_VALU --kernel-trace --output-format csv \
      -d $R/gpurun_out/sq_tc -- python3 $R/bench.py --no-cpu-baseline --no-roofline --no-extras --steps 3 --warmup 1 > $R/gpurun_out/sq_tc.log 2>&1
  python3 - "$(find $R/gpurun_out/sq_tc -name '*counter_collection.csv' | head -1)" "$(find $R/gpurun_out/sq_tc -name '*kernel_trace.csv' | head -1)" $lib <<'PY'
import csv, sys, collections
agg = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if "k_rollout_cartpole_mlp" in r["Kernel_Name"]:
        agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
dur = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-3 for r in csv.DictReader(open(sys.argv[2])) if "k_rollout_cartpole_mlp" in r["Kernel_Name"]]
print(sys.argv[3], "kernel_us_under_counters", round(sum(dur) / max(len(dur), 1), 2), {k: round(sum(v) / len(v)) for k, v in sorted(agg.items())})
PY
done
