#!/usr/bin/env python3
"""Summarise the rocprofv3 passes of tools/pmc_collect.sh into profiles/: per-kernel counter means
(`<tag>_pmc.csv`), kernel statistics (`<tag>_kernel_stats.csv`) and `pmc.json` (what bench.py reads for
roofline.traffic / roofline.valu).  HBM bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 (gfx950: FETCH_SIZE
counts half of a wide coalesced read; MI355X_MICROARCH.md)."""
import csv, glob, json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r04"
src = os.path.join(ROOT, "gpurun_out", f"prof_{tag}")
NAMES = {"blend2_fwd_batch_kernel": "blend_fwd_pair_kernel<40>",      # r04: the default pair forward (batched, fp16 pieces)
         "blend2_fwd_kernel<32, true, true, true": "blend_fwd_pair_kernel<40> (exact order)",
         "blend2_fwd_kernel<3,": "blend_fwd_kernel<3>", "blend2_fwd_kernel<8,": "blend_fwd_kernel<8>",
         "blend2_fwd_kernel<32,": "blend_fwd_kernel<32>", "blend2_bwd_narrow_kernel<3,": "blend_bwd_kernel<3>",
         "blend2_bwd_narrow_kernel<8,": "blend_bwd_kernel<8>", "blend2_bwd_wide_kernel<true, 0, 32, false, false": "blend_bwd_kernel<32>",
         "blend2_bwd_wide_kernel<true, 0, 32, false, true": "blend_bwd_pair_kernel<40>"}


def short(kernel):
    k = kernel.replace("void ", "")
    for frag, name in NAMES.items():
        if k.startswith(frag):
            return name
    return None


acc = {}
for path in glob.glob(os.path.join(src, "*", "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(path)):
        name = short(row.get("Kernel_Name", ""))
        if name is None:
            continue
        d = acc.setdefault((name, row["Counter_Name"]), [])
        d.append(float(row["Counter_Value"]))
# rocprofv3 writes one row per dispatch, counter and (for some counters) per dimension instance: sum the
# instances of one dispatch.  Dispatch ids are unique per pass.
acc2 = {}
for path in glob.glob(os.path.join(src, "*", "**", "*counter_collection.csv"), recursive=True):
    per = {}
    for row in csv.DictReader(open(path)):
        name = short(row.get("Kernel_Name", ""))
        if name is None:
            continue
        key = (name, row["Counter_Name"], row["Dispatch_Id"])
        per[key] = per.get(key, 0.0) + float(row["Counter_Value"])
    for (name, ctr, _), v in per.items():
        acc2.setdefault((name, ctr), []).append(v)
rows = sorted((k[0], k[1], sum(v) / len(v), len(v)) for k, v in acc2.items())
os.makedirs(os.path.join(ROOT, "profiles"), exist_ok=True)
with open(os.path.join(ROOT, "profiles", f"{tag}_pmc.csv"), "w") as f:
    f.write("kernel,counter,mean_value_per_dispatch,dispatches\n")
    for r in rows:
        f.write(f"\"{r[0]}\",{r[1]},{r[2]:.2f},{r[3]}\n")
mean = {(r[0], r[1]): r[2] for r in rows}
mix = json.loads(subprocess.check_output([sys.executable, os.path.join(ROOT, "tools", "isa_mix.py")]))
try:
    head = subprocess.check_output(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], text=True).strip()
    dirty = bool(subprocess.check_output(["git", "-C", ROOT, "status", "--porcelain", "--", "gaussiangrasper_amd/csrc",
                                          "include"], text=True).strip())
except Exception:
    head, dirty = "unknown", False
out = {"_commit": head + (" + uncommitted kernel changes" if dirty else ""),
       "_source": f"profiles/{tag}_pmc.csv (separate rocprofv3 --pmc passes over tools/kprobe.py: 1 M Gaussians, "
                  "1600x1200 bench view); hbm_bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024 per dispatch; "
                  "valu_issue_weight from tools/isa_mix.py (static mix of the loop bodies)"}
for name in sorted({r[0] for r in rows}):
    rec = {}
    if (name, "FETCH_SIZE") in mean and (name, "WRITE_SIZE") in mean:
        rec["fetch_kb"], rec["write_kb"] = mean[(name, "FETCH_SIZE")], mean[(name, "WRITE_SIZE")]
        rec["hbm_bytes"] = (2 * rec["fetch_kb"] + rec["write_kb"]) * 1024
    for ctr, key in (("SQ_INSTS_VALU", "insts_valu"), ("SQ_INSTS_SALU", "insts_salu"), ("SQ_INSTS_SMEM", "insts_smem"),
                     ("SQ_WAVE_CYCLES", "wave_cycles"), ("SQ_WAIT_ANY", "wait_any"), ("SQ_WAIT_INST_ANY", "wait_inst_any"),
                     ("SQ_ACTIVE_INST_ANY", "active_inst_any"), ("SQ_VALU_MFMA_BUSY_CYCLES", "mfma_busy_cycles"),
                     ("SQ_INSTS_LDS", "insts_lds"), ("SQ_LDS_BANK_CONFLICT", "lds_bank_conflict"),
                     ("VALUBusy", "valu_busy_pct"), ("SQ_VALU_MFMA_COEXEC_CYCLES", "valu_mfma_coexec_cycles"),
                     ("SQ_ACTIVE_INST_VALU", "active_inst_valu"), ("SQ_BUSY_CYCLES", "sq_busy_cycles"),
                     ("OccupancyPercent", "occupancy_pct"), ("TCC_EA0_ATOMIC_sum", "atomic_requests_64B"),
                     ("TCP_UTCL1_TRANSLATION_MISS_sum", "utcl1_misses")):
        if (name, ctr) in mean:
            rec[key] = mean[(name, ctr)]
    if name in mix:
        rec["valu_issue_weight"] = mix[name]["valu_issue_weight"]
        rec["loop_mix"] = mix[name]["by_class"]
    out[name] = rec
json.dump(out, open(os.path.join(ROOT, "profiles", "pmc.json"), "w"), indent=1)
for path in glob.glob(os.path.join(src, "stats", "**", "*kernel_stats.csv"), recursive=True):
    import shutil
    shutil.copy(path, os.path.join(ROOT, "profiles", f"{tag}_kernel_stats.csv"))
print(json.dumps(out, indent=1)[:3000])
