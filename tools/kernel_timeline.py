import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "k_rollout_cartpole_mlp_mix" in r["Kernel_Name"]]
mid = idx[len(idx)//2]; nxt = idx[len(idx)//2 + 1]
prev_end = None
for r in rows[mid:nxt+1]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = (s - prev_end) / 1e3 if prev_end else 0
    print(f"{r['Kernel_Name'].split('(')[0][-40:]:42s} dur {(e-s)/1e3:7.2f} us  gap before {gap:6.2f} us")
    prev_end = e
