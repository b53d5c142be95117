#!/bin/bash
# Randomised parity at depth: five seeds of tools/fuzz_parity.py side by side (the oracle is one CPU thread per process; a
# GPU box admits six processes on the card), each drawing cases until its time limit; tallies go to gpurun_out/fuzz_<seed>.txt
# and are added up in gpurun_out/fuzz_total.json.   usage: tools/fuzz_round.sh [seconds per process = 420] [first seed = 100]
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out
LIMIT=${1:-420}
SEED0=${2:-100}
pids=()
for k in 0 1 2 3 4; do
  s=$((SEED0 + k))
  python tools/fuzz_parity.py --cases 100000 --seed $s --time-limit $LIMIT --skip sharded_tail > gpurun_out/fuzz_$s.txt 2>&1 &
  pids+=($!)
done
for p in "${pids[@]}"; do wait $p; done
# the in-process ranks of the `sharded_tail` kind spin on each other's kernels: they need their hardware queues RESIDENT, which five
# processes side by side do not leave them (observed: dead waits until the time-out) -- one process, afterwards, a fifth of the time
python tools/fuzz_parity.py --cases 100000 --seed $((SEED0 + 5)) --time-limit $((LIMIT / 5)) --only sharded_tail > gpurun_out/fuzz_$((SEED0 + 5)).txt 2>&1
python - <<'PY'
import glob, json, os
tot, cases, ok, secs = {}, 0, True, 0.0
for f in sorted(glob.glob("gpurun_out/fuzz_[0-9]*.txt")):
    lines = [l for l in open(f) if l.startswith("{")]
    bad = [l for l in open(f) if l.startswith("MISMATCH")]
    if not lines:
        print("no tally in", f); ok = False; continue
    d = json.loads(lines[-1])
    cases += d["cases"]; ok = ok and d["all_bit_exact"] and not bad; secs = max(secs, d.get("seconds", 0))
    for k, v in d["by_kind"].items():
        t = tot.setdefault(k, {"cases": 0, "bit_exact": 0}); t["cases"] += v["cases"]; t["bit_exact"] += v["bit_exact"]
out = {"cases": cases, "processes": len(glob.glob("gpurun_out/fuzz_[0-9]*.txt")), "seconds": secs, "by_kind": tot, "all_bit_exact": ok}
json.dump(out, open("gpurun_out/fuzz_total.json", "w"))
print(json.dumps(out))
PY
