#!/usr/bin/env python3
"""Pair kernels (feature 32 | rgb + depth + normal 7) on the bench view, timed by the in-library hipEvent brackets:
forward and backward, with the quad lists (round 3) and without.  A/B runs of build variants:

    python tools/pairbench.py [--lib path/to/libvariant.so] [--reps 8] [--label text]

prints ONE line: label, fwd / bwd ms with lists, fwd / bwd ms without."""
import argparse
import ctypes
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "shim")]
import torch  # noqa: E402

from gaussiangrasper_amd import _lib, ops  # noqa: E402
from gaussiangrasper_amd.camera import ring_cameras  # noqa: E402
from gaussiangrasper_amd.scene import make_scene  # noqa: E402


def prof(lib):
    out = {}
    for kid in range(40):
        n, ms = ctypes.c_int(0), ctypes.c_double(0.0)
        if lib.gg_prof_get(kid, ctypes.byref(n), ctypes.byref(ms)) != 0:
            continue
        if n.value:
            out[lib.gg_prof_name(kid).decode()] = ms.value / n.value
    lib.gg_prof_reset()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--lib", default=None)
    ap.add_argument("--reps", type=int, default=8)
    ap.add_argument("--points", type=int, default=1_000_000)
    ap.add_argument("--label", default="")
    ap.add_argument("--view", type=int, default=0)
    ap.add_argument("--onesweep", type=int, default=None, help="gg_debug_set_depth_onesweep(0 / 1) before the runs")
    args = ap.parse_args()
    if args.lib:
        _lib.LIB_PATH = os.path.abspath(args.lib)
    lib = _lib.load(build_if_missing=False)
    if args.onesweep is not None:
        lib.gg_debug_set_depth_onesweep(args.onesweep)
    dev = "cuda:0"
    h, w = 1200, 1600
    sc = make_scene(args.points, config_index=3).to(dev)
    v = ring_cameras(8, h, w, device=dev)[args.view]
    xys, depths, radii, conics, nth, _ = ops.ProjectGaussians.apply(
        sc.means, sc.scales.exp(), 1, sc.quats, v.viewmat[:3], v.projmat, v.fx, v.fy, v.cx, v.cy, h, w,
        v.tile_bounds)
    opac = torch.sigmoid(sc.opacities)
    feat, tail = sc.feature, torch.rand(args.points, 7, device=dev)
    vo = [torch.randn(h, w, 32, device=dev), torch.randn(h, w, 7, device=dev)]
    res = {}
    for lists in (True, False):
        ops.USE_QUAD_LISTS = lists
        for phase in ("warm", "timed"):
            lib.gg_prof_reset()
            lib.gg_prof_enable(1 if phase == "timed" else 0)
            for _ in range(2 if phase == "warm" else args.reps):
                f = feat.detach().requires_grad_(True)
                t = tail.detach().requires_grad_(True)
                x = xys.detach().requires_grad_(True)
                ops.clear_bin_cache()
                imgs = ops.rasterize_segments(x, depths, radii, conics.detach(), nth, opac.detach(), h, w,
                                              [(f, torch.zeros(32, device=dev)), (t, torch.zeros(7, device=dev))])
                torch.autograd.backward(imgs, vo)
            torch.cuda.synchronize()
        lib.gg_prof_enable(0)
        r = prof(lib)
        res["lists" if lists else "walk"] = {k: round(t, 4) for k, t in r.items() if "pair" in k or "bin_sort" in k}
    print(json.dumps({"label": args.label or (args.lib or "product"), **res}), flush=True)


if __name__ == "__main__":
    main()
