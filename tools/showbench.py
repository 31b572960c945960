#!/usr/bin/env python3
"""Print the interesting fields of a bench.py JSON line (argument: log file)."""
import json, sys
d = json.loads([ln for ln in open(sys.argv[1]).read().splitlines() if ln.startswith("{")][-1])
print("value %.1f %s  ms/step %.2f  kernel-fraction %s  kernel-ms/view %s" % (
    d["value"], d["unit"], d["ms_per_step"], d.get("kernel_time_fraction_of_wall"), d.get("kernel_ms_per_view")))
print(" ".join(f"{k}={v}" for k, v in d["kernels"].items()))
r = d.get("roofline")
if r:
    print("dominant", r["kernel"], "frac %.4f" % r["frac"], "valu", r.get("valu"), "whole_view frac %.4f" % r["whole_view"]["frac"])
f = d.get("other_route") or d.get("fused_single_call_path") or {}
print("other route:", f.get("route"), f.get("value"), f.get("kernel_ms_per_view"), f.get("kernels"), "cpu", (d.get("cpu_baseline") or {}).get("value"))
