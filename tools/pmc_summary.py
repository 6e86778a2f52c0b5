#!/usr/bin/env python3
"""per-kernel mean of rocprofv3 --pmc counters:  pmc_summary.py counter_collection.csv [name-filter]"""
import csv, sys, collections
rows = csv.DictReader(open(sys.argv[1]))
filt = sys.argv[2] if len(sys.argv) > 2 else ""
acc = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
seen = set()
for r in rows:
    k = r["Kernel_Name"]
    if filt and filt not in k:
        continue
    if "Grid_Size" in r:                       # same kernel, different launch sizes (layers): one line each
        k = k[:44] + " g" + str(r["Grid_Size"])
    acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    key = (k, r["Dispatch_Id"])
    if key not in seen:
        seen.add(key)
        dur[k].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
names = sorted({c for k in acc for c in acc[k]})
print("kernel".ljust(50), "n".rjust(4), "us".rjust(8), *[n.replace("SQ_", "")[:14].rjust(15) for n in names])
for k in sorted(acc, key=lambda k: -sum(dur[k])):
    vals = [sum(acc[k][n]) / max(len(acc[k][n]), 1) for n in names]
    print(k[-50:].ljust(50) if " g" in k[-12:] else k[:50].ljust(50), str(len(dur[k])).rjust(4), f"{sum(dur[k]) / len(dur[k]):8.1f}", *[f"{v:15.3e}" for v in vals])
