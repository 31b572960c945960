#!/usr/bin/env python3
"""Forward walks over several 32-channel blocks (gg_debug_set_fwd_blocks) on BASELINE config 5's view: 5 M Gaussians,
1920x1080, 128-channel feature + rgb | depth | normal, render-only.  One line per (pair blocks, chunk blocks) setting:
forward kernel times of the view (in-library hipEvents) and whether the images are bit-identical to the 1/1 setting.
    python tools/fwdblocks_bench.py [--points 5000000] [--settings 1,1 2,2 4,1 1,3 ...]"""
import argparse, ctypes, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "shim")]
import torch
from gaussiangrasper_amd import _lib, ops
from gaussiangrasper_amd.camera import ring_cameras
from gaussiangrasper_amd.scene import make_scene

ap = argparse.ArgumentParser()
ap.add_argument("--points", type=int, default=5_000_000)
ap.add_argument("--feature-dim", type=int, default=128)
ap.add_argument("--settings", nargs="*", default=["1,1", "1,3", "2,2", "2,1", "4,1", "1,2"])
a = ap.parse_args()
lib = _lib.load(build_if_missing=False)
dev, h, w = "cuda:0", 1080, 1920
sc = make_scene(a.points, feature_dim=a.feature_dim, config_index=4).to(dev)
v = ring_cameras(8, h, w, device=dev)[0]
with torch.no_grad():
    xys, depths, radii, conics, nth, _ = ops.ProjectGaussians.apply(
        sc.means, sc.scales.exp(), 1, sc.quats, v.viewmat[:3], v.projmat, v.fx, v.fy, v.cx, v.cy, h, w, v.tile_bounds)
    opac = torch.sigmoid(sc.opacities)
    feat, tail = sc.feature.detach(), torch.rand(a.points, 7, device=dev)
    segs = [(feat, torch.zeros(a.feature_dim, device=dev)), (tail, torch.zeros(7, device=dev))]
    ref = None
    for st in a.settings:
        pb, cb = (int(t) for t in st.split(","))
        lib.gg_debug_set_fwd_blocks(pb, cb)
        for phase in ("warm", "timed"):
            lib.gg_prof_reset()
            lib.gg_prof_enable(1 if phase == "timed" else 0)
            for _ in range(3):
                imgs = ops.rasterize_segments(xys, depths, radii, conics, nth, opac, h, w, segs)
            torch.cuda.synchronize()
        lib.gg_prof_enable(0)
        t = {}
        for kid in range(40):
            n, ms = ctypes.c_int(0), ctypes.c_double(0.0)
            if lib.gg_prof_get(kid, ctypes.byref(n), ctypes.byref(ms)) == 0 and n.value:
                name = lib.gg_prof_name(kid).decode()
                if "blend_fwd" in name:
                    t[name] = round(ms.value / 3, 4)      # per view (all launches of that kernel)
        same = None
        if ref is None:
            ref = [i.clone() for i in imgs]
        else:
            same = all(torch.equal(x, y) for x, y in zip(imgs, ref))
        print(json.dumps({"pair_blocks": pb, "chunk_blocks": cb, "fwd_ms_per_view": round(sum(t.values()), 4), "kernels": t,
                          "bit_identical_to_first": same}))
lib.gg_debug_set_fwd_blocks(1, 1)
