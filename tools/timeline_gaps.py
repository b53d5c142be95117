import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
key = sys.argv[2] if len(sys.argv) > 2 else "k_rollout_cartpole_mlp"
idx = [i for i, r in enumerate(rows) if key in r["Kernel_Name"]]
gaps, durs = collections.defaultdict(list), collections.defaultdict(list)
lo = len(idx) // 4
for a in range(lo, min(lo + 200, len(idx) - 1)):
    prev_end = None
    for r in rows[idx[a]:idx[a + 1] + 1]:
        name = r["Kernel_Name"].split("(")[0][-40:]
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        side = "init_states" in name
        if prev_end is not None and not side:
            gaps[name].append((s - prev_end) / 1e3)
        durs[name].append((e - s) / 1e3)
        if not side:
            prev_end = e
for k in durs:
    print(f"{k:42s} dur {sum(durs[k]) / len(durs[k]):7.2f} us  gap before {sum(gaps[k]) / max(len(gaps[k]), 1):6.2f} us  n={len(durs[k])}")
