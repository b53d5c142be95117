#!/bin/bash
# HBM traffic counters for the bench's kernels: two separate rocprofv3 --pmc passes (FETCH_SIZE needs 3 of the
# 4 TCC slots, WRITE_SIZE 2 -- MI355X_MICROARCH.md "rocprofv3 PMC slots"), kernel-trace only.
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf $R/gpurun_out/pmc_$c
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $R/gpurun_out/pmc_$c -- python3 $R/bench.py --no-cpu-baseline --no-extras --blocks 1 --steps 5 --warmup 1 > $R/gpurun_out/pmc_$c.log 2>&1
  f=$(find $R/gpurun_out/pmc_$c -name "*counter_collection.csv" | head -1)
  echo "== $c ($f)"
  python3 - "$f" "$c" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(list)
for r in rows:
    if r.get("Counter_Name") == sys.argv[2]:
        agg[r["Kernel_Name"].split("(")[0][:70]].append(float(r["Counter_Value"]))
for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1]))[:8]:
    print(f"{k:72s} n={len(v):3d} avg={sum(v)/len(v):14.1f} (counter units)")
PY
done
