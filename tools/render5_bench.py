#!/usr/bin/env python3
"""BASELINE config 5 timing: 5 M Gaussians, 128-dim feature, 1920x1080, render-only (project +
bin/sort + 128-channel forward), views/s on one GPU.  Prints one JSON line."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "shim")]
import torch
from gaussiangrasper_amd import ops
from gaussiangrasper_amd.camera import ring_cameras
from gaussiangrasper_amd.scene import make_scene

dev = "cuda:0"
n, h, w, d = 5_000_000, 1080, 1920, 128
sc = make_scene(n, feature_dim=d, sh_degree=0, config_index=4)
views = ring_cameras(8, h, w, device=dev)
means, scales, quats = sc.means.to(dev), sc.scales.exp().to(dev), sc.quats.to(dev)
opac, feat = torch.sigmoid(sc.opacities.to(dev)), sc.feature.to(dev)
bg = torch.zeros(d, device=dev)

def render(v):
    xys, depths, radii, conics, nth, _ = ops.ProjectGaussians.apply(
        means, scales, 1, quats, v.viewmat[:3], v.projmat, v.fx, v.fy, v.cx, v.cy, h, w, v.tile_bounds)
    return ops.NDRasterizeGaussians.apply(xys, depths, radii, conics, nth, feat, opac, h, w, bg), nth

with torch.no_grad():
    _, nth = render(views[0])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for rep in range(2):
        for v in views:
            render(v)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
print(json.dumps({"workload": "BASELINE config 5 per-GPU share: 5M Gaussians, 128-ch feature, 1920x1080, render-only",
                  "views_per_s": 16 / dt, "ms_per_view": 1e3 * dt / 16, "num_intersects_view0": int(nth.long().sum())}))
