#!/usr/bin/env python3
"""Pair kernels (feature 32 | rgb + depth + normal 7) and the binning on the bench view, timed by the in-library
hipEvent brackets.  A/B runs of build variants, interleaved in ONE process (separate invocations add cross-process
and cross-device variance that looks like a kernel property):

    python tools/pairbench.py [--libs product variants/libA.so variants/libB.so] [--rounds 3] [--reps 6]

prints one JSON line per library: label, per-kernel ms (median over the rounds, and the minimum)."""
import argparse
import ctypes
import json
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "shim")]
import torch  # noqa: E402

from gaussiangrasper_amd import _lib, ops  # noqa: E402
from gaussiangrasper_amd.camera import ring_cameras  # noqa: E402
from gaussiangrasper_amd.scene import make_scene  # noqa: E402


def prof(lib):
    out = {}
    for kid in range(48):
        n, ms = ctypes.c_int(0), ctypes.c_double(0.0)
        if lib.gg_prof_get(kid, ctypes.byref(n), ctypes.byref(ms)) != 0:
            continue
        if n.value:
            out[lib.gg_prof_name(kid).decode()] = ms.value / n.value
    lib.gg_prof_reset()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--libs", nargs="*", default=["product"])
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--reps", type=int, default=6)
    ap.add_argument("--points", type=int, default=1_000_000)
    ap.add_argument("--view", type=int, default=0)
    ap.add_argument("--fast-forward", type=int, default=None, help="ops.set_fast_forward(0 / 1) before the runs")
    ap.add_argument("--keys", default="pair,bin_sort", help="substrings of the kernel names to report")
    args = ap.parse_args()
    product = _lib.load(build_if_missing=False)
    libs = [(p, product if p == "product" else _lib.load_variant(os.path.abspath(p))) for p in args.libs]
    if args.fast_forward is not None and hasattr(ops, "set_fast_forward"):
        ops.set_fast_forward(bool(args.fast_forward))
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
    keys = [k for k in args.keys.split(",") if k]

    def run(lib, reps, timed):
        _lib._lib = lib                      # the operators call whatever _lib.load() returns
        lib.gg_prof_reset()
        lib.gg_prof_enable(1 if timed else 0)
        for _ in range(reps):
            f = feat.detach().requires_grad_(True)
            t = tail.detach().requires_grad_(True)
            x = xys.detach().requires_grad_(True)
            ops.clear_bin_cache()
            imgs = ops.rasterize_segments(x, depths, radii, conics.detach(), nth, opac.detach(), h, w,
                                          [(f, torch.zeros(32, device=dev)), (t, torch.zeros(7, device=dev))])
            torch.autograd.backward(imgs, vo)
        torch.cuda.synchronize()
        lib.gg_prof_enable(0)
        return prof(lib) if timed else None

    res = {p: [] for p, _ in libs}
    try:
        for p, lib in libs:
            run(lib, 2, False)
        for _ in range(args.rounds):
            for p, lib in libs:
                res[p].append(run(lib, args.reps, True))
    finally:
        _lib._lib = product
    for p, _ in libs:
        names = sorted({k for r in res[p] for k in r if any(s in k for s in keys)})
        med = {k: round(statistics.median(r[k] for r in res[p] if k in r), 4) for k in names}
        mn = {k: round(min(r[k] for r in res[p] if k in r), 4) for k in names}
        print(json.dumps({"label": p, "median_ms": med, "min_ms": mn}), flush=True)


if __name__ == "__main__":
    main()
