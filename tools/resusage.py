#!/usr/bin/env python3
"""Summarise hipcc's -Rpass-analysis=kernel-resource-usage remarks: one line per kernel.
Usage: hipcc ... -Rpass-analysis=kernel-resource-usage 2> res.txt ; python tools/resusage.py res.txt [filter]"""
import re
import subprocess
import sys

txt = open(sys.argv[1]).read()
flt = sys.argv[2] if len(sys.argv) > 2 else ""
cur = None
rows = {}
for line in txt.splitlines():
    m = re.search(r"Function Name: (\S+)", line)
    if m:
        cur = m.group(1)
        rows[cur] = {}
        continue
    m = re.search(r"remark:\s+([A-Za-z ]+?)(?: \[[a-zA-Z/]+\])?: (\d+)", line)
    if m and cur:
        rows[cur][m.group(1).strip()] = int(m.group(2))
names = list(rows)
dem = subprocess.run(["c++filt"] + names, capture_output=True, text=True).stdout.splitlines()
for n, d in zip(names, dem):
    d = re.sub(r"\(.*", "", d).replace("void ", "")
    if flt and flt not in d:
        continue
    r = rows[n]
    print(f"{d:62s} VGPR {r.get('VGPRs', -1):4d} AGPR {r.get('AGPRs', -1):3d} spill {r.get('VGPRs Spill', -1):3d} "
          f"scratch {r.get('ScratchSize', -1):4d} occ {r.get('Occupancy', -1)} LDS {r.get('LDS Size', -1)}")
