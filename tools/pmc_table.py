#!/usr/bin/env python3
"""Per-kernel means of the counters collected by tools/pmc_kernel.sh: python tools/pmc_table.py gpurun_out/pmc_<tag> [filter]"""
import csv, glob, os, sys
src = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else "blend2"
acc = {}
for path in glob.glob(os.path.join(src, "*", "**", "*counter_collection.csv"), recursive=True):
    per = {}
    for row in csv.DictReader(open(path)):
        k = row.get("Kernel_Name", "")
        if flt not in k:
            continue
        key = (k.split("(")[0][:70], row["Counter_Name"], row["Dispatch_Id"])
        per[key] = per.get(key, 0.0) + float(row["Counter_Value"])
    for (k, c, _), v in per.items():
        acc.setdefault((k, c), []).append(v)
names = sorted({k for k, _ in acc})
for k in names:
    print(k)
    for (kk, c), v in sorted(acc.items()):
        if kk == k:
            print("    %-28s %16.1f   (%d dispatches)" % (c, sum(v) / len(v), len(v)))
