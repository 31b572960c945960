#!/usr/bin/env python3
"""Experiment (measurement twin of the library): blend kernel times on the bench view with the default
workgroup -> tile mapping (strips of 4 tiles dealt to the XCDs) against longest-list-first orders."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "shim"), os.path.join(ROOT, "tools")]
import torch
from gaussiangrasper_amd import _lib, build as gg_build, ops
from gaussiangrasper_amd.camera import ring_cameras
from gaussiangrasper_amd.scene import make_scene
from kbench import prof

dev = "cuda:0"
gg_build.build_ablation()
_lib.LIB_PATH = gg_build.ABL_OUT
lib = _lib.load()
lib.gg_debug_set_tile_order.restype, lib.gg_debug_set_tile_order.argtypes = ctypes.c_int, [ctypes.c_void_p]
h, w = 1200, 1600
n = 1_000_000
sc = make_scene(n, config_index=3).to(dev)
v = ring_cameras(8, h, w, device=dev)[0]
xys, depths, radii, conics, nth, _ = ops.ProjectGaussians.apply(
    sc.means, sc.scales.exp(), 1, sc.quats, v.viewmat[:3], v.projmat, v.fx, v.fy, v.cx, v.cy, h, w,
    v.tile_bounds)
opac = torch.sigmoid(sc.opacities)
rgb = torch.rand(n, 3, device=dev)
tail = torch.rand(n, 7, device=dev)
feat = sc.feature.detach()
b = ops.bin_and_sort_gaussians(xys, depths, radii, nth, h, w)
lens = (b.tile_bins[:, 1] - b.tile_bins[:, 0]).long()
ntiles = lens.numel()
tx = v.tile_bounds[0]


def run(label):
    res = {}
    for cols, op in ((rgb, ops.RasterizeGaussians), (feat, ops.NDRasterizeGaussians)):
        c = cols.detach().requires_grad_(True)
        x = xys.detach().requires_grad_(True)
        vo = torch.randn(h, w, c.shape[1], device=dev)
        bg = torch.zeros(c.shape[1], device=dev)
        for rep in range(2):
            lib.gg_prof_reset(); lib.gg_prof_enable(1)
            for _ in range(4):
                out = op.apply(x, depths, radii, conics.detach(), nth, c, opac.detach(), h, w, bg)
                out.backward(vo)
            torch.cuda.synchronize(); lib.gg_prof_enable(0)
            r = prof(lib)
        res.update({k: round(t, 4) for k, t in r.items() if "blend_" in k and "prep" not in k})
    # the plugin route's pair kernels
    f = feat.detach().requires_grad_(True)
    t7 = tail.detach().requires_grad_(True)
    x = xys.detach().requires_grad_(True)
    vo32, vo7 = torch.randn(h, w, 32, device=dev), torch.randn(h, w, 7, device=dev)
    for rep in range(2):
        lib.gg_prof_reset(); lib.gg_prof_enable(1)
        for _ in range(4):
            outs = ops.rasterize_segments(x, depths, radii, conics.detach(), nth, opac.detach(), h, w,
                                          [(f, torch.zeros(32, device=dev)), (t7, torch.zeros(7, device=dev))])
            torch.autograd.backward(outs, [vo32, vo7])
        torch.cuda.synchronize(); lib.gg_prof_enable(0)
        r = prof(lib)
    res.update({k: round(t, 4) for k, t in r.items() if "pair" in k})
    print(label, res, flush=True)


run("default order      ")
order = torch.argsort(lens, descending=True, stable=True).int().contiguous()
lib.gg_debug_set_tile_order(order.data_ptr())
run("longest first      ")
# longest first in coarse classes (log2 of the length), original strip order inside a class: keeps neighbours
cls = torch.floor(torch.log2(lens.clamp(min=1).float())).long()
order2 = torch.argsort(-cls, stable=True).int().contiguous()
lib.gg_debug_set_tile_order(order2.data_ptr())
run("log2 classes, desc ")
# plain row-major (no XCD strips) for reference
order3 = torch.arange(ntiles, device=dev).int().contiguous()
lib.gg_debug_set_tile_order(order3.data_ptr())
run("row-major          ")
lib.gg_debug_set_tile_order(None)
run("default again      ")
