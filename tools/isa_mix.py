#!/usr/bin/env python3
"""Static VALU instruction mix of the blend kernels' loops (for roofline.valu.issue_weight).

Compiles csrc/blend2.hip to gfx950 assembly and, per kernel, counts the vector-ALU instructions between
the first loop header and the last backward branch, by issue-cost class:
    plain wave64 VALU                       1      (2 cycles on a SIMD-32; MI355X_MICROARCH.md)
    DPP (quad_perm / row_* modifiers)       2.5    (tools/ubench_xlane.hip: 2.5 ns vs 0.98 ns)
    v_permlane16/32_swap                    5.5    (tools/ubench_xlane.hip)
    transcendental (rcp, exp, log, sqrt..)  4      (8 cycles; MI355X_MICROARCH.md cycle constants)
    packed fp32 (v_pk_*_f32)                2
MFMA instructions run on the matrix pipe and are not VALU issue.  The weight is the cost-weighted count
divided by the plain count: SQ_INSTS_VALU x weight x 2 cycles = VALU issue cycles of the kernel.
A static mix of the loop bodies (not a dynamic one): the loops are almost branch-free, so it is close."""
import json, os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "gaussiangrasper_amd", "csrc", "blend2.hip")
KERNELS = {  # bench.py name -> mangled-name fragment
    "blend_fwd_kernel<3>": "blend2_fwd_kernelILi3ELb0",
    "blend_fwd_kernel<8>": "blend2_fwd_kernelILi8ELb0",
    "blend_fwd_kernel<32>": "blend2_fwd_kernelILi32ELb1ELb1ELb0E",
    "blend_fwd_pair_kernel<40>": "blend2_fwd_batch_kernel",
    "blend_fwd_pair_kernel<40> (exact order)": "blend2_fwd_kernelILi32ELb1ELb1ELb1E",
    "blend_bwd_kernel<3>": "blend2_bwd_narrow_kernelILi3ELi0ELb0E",
    "blend_bwd_kernel<8>": "blend2_bwd_narrow_kernelILi8ELi0ELb0E",
    "blend_bwd_kernel<32>": "blend2_bwd_wide_kernelILb1ELi0ELi32ELb0ELb0E",
    "blend_bwd_pair_kernel<40>": "blend2_bwd_wide_kernelILb1ELi0ELi32ELb0ELb1E",
}
TRANS = ("v_rcp_", "v_exp_", "v_log_", "v_sqrt_", "v_rsq_", "v_sin_", "v_cos_")


def classify(line):
    op = line.split()[0]
    if not op.startswith("v_") or op.startswith("v_mfma"):
        return None
    if op.startswith("v_permlane"):
        return "permlane"
    if "quad_perm" in line or "row_" in line or "_dpp" in op or "wave_" in line:
        return "dpp"
    if op.startswith(TRANS):
        return "trans"
    if op.startswith("v_pk_") and "f32" in op:
        return "pk"
    return "plain"


COST = {"plain": 1.0, "dpp": 2.5, "permlane": 5.5, "trans": 4.0, "pk": 2.0}


def main():
    with tempfile.TemporaryDirectory() as td:
        asm = os.path.join(td, "blend2.s")
        sys.path.insert(0, ROOT)
        from gaussiangrasper_amd.build import FLAGS     # the product's flags (minus the link step)
        flags = [f for f in FLAGS if f not in ("-shared", "-fPIC")]
        subprocess.check_call(["/opt/rocm/bin/hipcc", *flags, "-S", "--cuda-device-only", "-o", asm, SRC],
                              stderr=subprocess.DEVNULL)
        text = open(asm).read().splitlines()
    out = {}
    for name, frag in KERNELS.items():
        start = next((i for i, l in enumerate(text) if l.startswith("_Z") and frag in l and l.rstrip().endswith(":") or
                      (l.startswith("_Z") and frag in l and ":" in l and "@" in l)), None)
        if start is None:
            continue
        end = next(i for i in range(start + 1, len(text)) if text[i].strip().startswith("s_endpgm"))
        body = text[start:end]
        labels = {l.split(":")[0].strip(): i for i, l in enumerate(body) if re.match(r"^\.LBB\d+_\d+:", l)}
        first_loop = next((i for i, l in enumerate(body) if "Loop Header" in l), 0)
        last_back = 0
        for i, l in enumerate(body):
            m = re.search(r"s_cbranch\w*\s+(\.LBB\d+_\d+)", l) or re.search(r"s_branch\s+(\.LBB\d+_\d+)", l)
            if m and labels.get(m.group(1), 10 ** 9) < i:
                last_back = max(last_back, i)
        counts = {k: 0 for k in COST}
        for l in body[first_loop:last_back + 1]:
            ls = l.strip()
            if not ls or ls.startswith((";", ".", "//")):
                continue
            c = classify(ls)
            if c:
                counts[c] += 1
        total = sum(counts.values())
        weighted = sum(COST[k] * v for k, v in counts.items())
        out[name] = {"loop_valu_instructions": total, "by_class": counts,
                     "valu_issue_weight": round(weighted / max(total, 1), 4)}
    json.dump(out, sys.stdout, indent=1)
    print()


if __name__ == "__main__":
    main()
