#!/usr/bin/env python3
"""Kernel-level measurement harness (GPU box): times the blend kernels on the bench workload
through the C ABI with the in-library hipEvent brackets, optionally with the ablated builds of the
3-channel backward.  Usage: python tools/kbench.py [--reps 5]"""
import argparse
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "shim")]
import torch  # noqa: E402

from gaussiangrasper_amd import _lib, build as gg_build, ops  # noqa: E402
from gaussiangrasper_amd.camera import ring_cameras  # noqa: E402
from gaussiangrasper_amd.scene import make_scene  # noqa: E402


def prof(lib):
    out = {}
    for kid in range(40):
        n, ms = ctypes.c_int(0), ctypes.c_double(0.0)
        lib.gg_prof_get(kid, ctypes.byref(n), ctypes.byref(ms))
        if n.value:
            out[lib.gg_prof_name(kid).decode()] = ms.value / n.value
    lib.gg_prof_reset()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--points", type=int, default=1_000_000)
    ap.add_argument("--skip-narrow", action="store_true")
    args = ap.parse_args()
    dev = "cuda:0"
    # the ablated kernels live only in the measurement twin of the library (-DGG_ABLATION)
    gg_build.build_ablation()       # always: tools/walkstats.py leaves a twin with the (slow) walk counters
    _lib.LIB_PATH = gg_build.ABL_OUT
    lib = _lib.load()
    lib.gg_debug_set_ablation.restype, lib.gg_debug_set_ablation.argtypes = ctypes.c_int, [ctypes.c_int]
    occ = (ctypes.c_int * 4)()
    if lib.gg_debug_occupancy(occ) == 0:
        print("resident workgroups per CU: bwd<32> %d, bwd pair %d, bwd<8> %d, fwd pair %d" % tuple(occ), flush=True)
    h, w = 1200, 1600
    sc = make_scene(args.points, config_index=3).to(dev)
    v = ring_cameras(8, h, w, device=dev)[0]
    xys, depths, radii, conics, nth, _ = ops.ProjectGaussians.apply(
        sc.means, sc.scales.exp(), 1, sc.quats, v.viewmat[:3], v.projmat, v.fx, v.fy, v.cx, v.cy, h, w,
        v.tile_bounds)
    opac = torch.sigmoid(sc.opacities)
    rgb = torch.rand(args.points, 3, device=dev)
    feat = sc.feature

    def run(colors, label):
        c = colors.detach().requires_grad_(True)
        x = xys.detach().requires_grad_(True)
        op = ops.RasterizeGaussians if c.shape[1] == 3 else ops.NDRasterizeGaussians
        vo = torch.randn(h, w, c.shape[1], device=dev)
        lib.gg_prof_reset()
        lib.gg_prof_enable(1)
        for _ in range(args.reps):
            out = op.apply(x, depths, radii, conics.detach(), nth, c, opac.detach(), h, w,
                           torch.zeros(c.shape[1], device=dev))
            out.backward(vo)
        torch.cuda.synchronize()
        lib.gg_prof_enable(0)
        r = prof(lib)
        print(label, {k: round(t, 4) for k, t in r.items() if "blend" in k}, flush=True)

    if not args.skip_narrow:
        run(rgb, "C=3 full   ")
        for lvl, name in ((1, "no atomics "), (2, "no butterfly"), (3, "geometry only"), (4, "staging only")):
            lib.gg_debug_set_ablation(lvl)
            run(rgb, f"C=3 abl{lvl} {name}")
        lib.gg_debug_set_ablation(0)
    run(feat, "C=32 full  ")
    for lvl, name in ((11, "no colour atomics"), (12, "no MFMA flush"), (13, "no butterfly"), (14, "no D product"),
                      (15, "no recurrence"), (16, "staging + queue only")):
        lib.gg_debug_set_ablation(lvl)
        run(feat, f"C=32 abl{lvl} {name}")
    lib.gg_debug_set_ablation(0)

    # the plugin route's operator: feature (32) | rgb + depth + normal (7), pair kernels forward and backward
    tail = torch.rand(args.points, 7, device=dev)

    def run_pair(label):
        f = feat.detach().requires_grad_(True)
        t = tail.detach().requires_grad_(True)
        x = xys.detach().requires_grad_(True)
        vo = [torch.randn(h, w, 32, device=dev), torch.randn(h, w, 7, device=dev)]
        lib.gg_prof_reset()
        lib.gg_prof_enable(1)
        for _ in range(args.reps):
            imgs = ops.rasterize_segments(x, depths, radii, conics.detach(), nth, opac.detach(), h, w,
                                          [(f, torch.zeros(32, device=dev)), (t, torch.zeros(7, device=dev))])
            torch.autograd.backward(imgs, vo)
        torch.cuda.synchronize()
        lib.gg_prof_enable(0)
        r = prof(lib)
        print(label, {k: round(v, 4) for k, v in r.items() if "bwd" in k and "blend" in k}, flush=True)

    def run_pair_fwd(label):
        f, t = feat.detach(), tail.detach()
        lib.gg_prof_reset()
        lib.gg_prof_enable(1)
        with torch.no_grad():
            for _ in range(args.reps):
                ops.clear_bin_cache()
                ops.rasterize_segments(xys.detach(), depths, radii, conics.detach(), nth, opac.detach(), h, w,
                                       [(f, torch.zeros(32, device=dev)), (t, torch.zeros(7, device=dev))])
        torch.cuda.synchronize()
        lib.gg_prof_enable(0)
        r = prof(lib)
        print(label, {k: round(v, 4) for k, v in r.items() if "fwd_pair" in k}, flush=True)

    lib.gg_debug_set_fwd_ablation.restype, lib.gg_debug_set_fwd_ablation.argtypes = ctypes.c_int, [ctypes.c_int]
    run_pair_fwd("fwd pair full")
    for lvl, name in ((1, "no MFMAs"), (2, "no colour-row loads"), (3, "no second-array fma"), (4, "staging only")):
        lib.gg_debug_set_fwd_ablation(lvl)
        run_pair_fwd(f"fwd pair abl{lvl} {name}")
    lib.gg_debug_set_fwd_ablation(0)

    run_pair("pair full  ")
    for lvl, name in ((1, "no colour atomics"), (2, "no flushes"), (3, "no butterfly"), (4, "no D product"),
                      (5, "no recurrence"), (6, "staging + queue only")):
        lib.gg_debug_set_pair_ablation(lvl)
        run_pair(f"pair abl{lvl} {name}")
    lib.gg_debug_set_pair_ablation(0)


if __name__ == "__main__":
    main()
