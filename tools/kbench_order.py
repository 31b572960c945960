#!/usr/bin/env python3
"""Kernel timing on the bench workload (in-library hipEvents): the 3- and 32-channel rasterize
forward/backward and the binning, two rounds of four launches each (the second round is reported).
`run(label, idx)` takes a permutation of the Gaussians: profiles/README.md records the experiment
that used it (Gaussians permuted into centre-tile order: no change)."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "shim"), os.path.join(ROOT, "tools")]
import torch
from gaussiangrasper_amd import _lib, ops
from gaussiangrasper_amd.camera import ring_cameras
from gaussiangrasper_amd.scene import make_scene
from kbench import prof

dev = "cuda:0"
lib = _lib.load()
h, w = 1200, 1600
n = 1_000_000
sc = make_scene(n, config_index=3).to(dev)
v = ring_cameras(8, h, w, device=dev)[0]
xys, depths, radii, conics, nth, _ = ops.ProjectGaussians.apply(
    sc.means, sc.scales.exp(), 1, sc.quats, v.viewmat[:3], v.projmat, v.fx, v.fy, v.cx, v.cy, h, w,
    v.tile_bounds)
opac = torch.sigmoid(sc.opacities)
rgb = torch.rand(n, 3, device=dev)
feat = sc.feature.detach()
tx = v.tile_bounds[0]
tile = (xys[:, 1].clamp(0, h - 1) // 16).long() * tx + (xys[:, 0].clamp(0, w - 1) // 16).long()
perm = torch.argsort(tile, stable=True)

def run(label, idx):
    a = [t[idx].contiguous() for t in (xys, depths, radii, conics, nth, opac)]
    for cols, op in ((rgb[idx].contiguous(), ops.RasterizeGaussians), (feat[idx].contiguous(), ops.NDRasterizeGaussians)):
        c = cols.detach().requires_grad_(True)
        x = a[0].detach().requires_grad_(True)
        vo = torch.randn(h, w, c.shape[1], device=dev)
        bg = torch.zeros(c.shape[1], device=dev)
        for rep in range(2):
            lib.gg_prof_reset(); lib.gg_prof_enable(1)
            for _ in range(4):
                ops.clear_bin_cache()
                out = op.apply(x, a[1], a[2], a[3].detach(), a[4], c, a[5].detach(), h, w, bg)
                out.backward(vo)
            torch.cuda.synchronize(); lib.gg_prof_enable(0)
            r = prof(lib)
        print(label, {k: round(t, 4) for k, t in r.items() if "blend" in k or "bin" in k}, flush=True)

ident = torch.arange(n, device=dev)
run(os.environ.get("KB_LABEL", "run") + " ", ident)
