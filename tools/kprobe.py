#!/usr/bin/env python3
"""Two launches of each blend kernel on the bench workload (for rocprofv3 --pmc passes)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "shim")]
import torch
from gaussiangrasper_amd import ops
from gaussiangrasper_amd.camera import ring_cameras
from gaussiangrasper_amd.scene import make_scene
dev = "cuda:0"
h, w, n = 1200, 1600, 1_000_000
sc = make_scene(n, config_index=3).to(dev)
v = ring_cameras(8, h, w, device=dev)[0]
xys, depths, radii, conics, nth, _ = ops.ProjectGaussians.apply(
    sc.means, sc.scales.exp(), 1, sc.quats, v.viewmat[:3], v.projmat, v.fx, v.fy, v.cx, v.cy, h, w, v.tile_bounds)
opac = torch.sigmoid(sc.opacities)
for cols, op in ((torch.rand(n, 3, device=dev), ops.RasterizeGaussians), (sc.feature.detach(), ops.NDRasterizeGaussians)):
    c = cols.requires_grad_(True)
    x = xys.detach().requires_grad_(True)
    vo = torch.randn(h, w, c.shape[1], device=dev)
    for _ in range(2):
        out = op.apply(x, depths, radii, conics.detach(), nth, c, opac.detach(), h, w, torch.zeros(c.shape[1], device=dev))
        out.backward(vo)
# the plugin route's operator: feature (32) | rgb + depth + normal (7) -> forward pair kernel, 8-channel
# rider backward, 32-channel backward
feat = sc.feature.detach().requires_grad_(True)
tail = torch.rand(n, 7, device=dev).requires_grad_(True)
x = xys.detach().requires_grad_(True)
vo = [torch.randn(h, w, 32, device=dev), torch.randn(h, w, 7, device=dev)]
for _ in range(2):
    imgs = ops.rasterize_segments(x, depths, radii, conics.detach(), nth, opac.detach(), h, w,
                                  [(feat, torch.zeros(32, device=dev)), (tail, torch.zeros(7, device=dev))])
    torch.autograd.backward(imgs, vo)
torch.cuda.synchronize()
# the exact-order pair forward (ops.set_exact_forward: the parity suite's kernel) beside the default one
ops.set_exact_forward(True)
with torch.no_grad():
    for _ in range(2):
        ops.rasterize_segments(x, depths, radii, conics.detach(), nth, opac.detach(), h, w,
                               [(feat, torch.zeros(32, device=dev)), (tail, torch.zeros(7, device=dev))])
torch.cuda.synchronize()
