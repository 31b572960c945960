#!/usr/bin/env python3
"""Walk statistics of the 3-channel forward on the bench view (measurement twin of the library,
-DGG_ABLATION): how many list entries are staged, survive the quadrant cull, have at least one blending
pixel; how many (pixel, Gaussian) pairs are evaluated / pass the alpha test / blend."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "shim")]
import torch
from gaussiangrasper_amd import _lib, build as gg_build, ops
from gaussiangrasper_amd.camera import ring_cameras
from gaussiangrasper_amd.scene import make_scene
gg_build.build_ablation(walk_stats=True)
_lib.LIB_PATH = gg_build.ABL_OUT
lib = _lib.load()
dev = "cuda:0"
h, w, n = 1200, 1600, int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
sc = make_scene(n, config_index=3).to(dev)
for vi in (0, 3):
    v = ring_cameras(8, h, w, device=dev)[vi]
    xys, depths, radii, conics, nth, _ = ops.ProjectGaussians.apply(
        sc.means, sc.scales.exp(), 1, sc.quats, v.viewmat[:3], v.projmat, v.fx, v.fy, v.cx, v.cy, h, w, v.tile_bounds)
    opac = torch.sigmoid(sc.opacities)
    torch.cuda.synchronize()
    buf = (ctypes.c_ulonglong * 8)()
    lib.gg_debug_walk_stats(None, 1)
    out = ops.RasterizeGaussians.apply(xys, depths, radii, conics, nth, torch.rand(n, 3, device=dev), opac, h, w,
                                       torch.zeros(3, device=dev))
    torch.cuda.synchronize()
    lib.gg_debug_walk_stats(buf, 0)
    s = list(buf)
    I = int(nth.long().sum())
    print(f"view {vi}: I={I} staged(x4 quadrants)={s[0]} walked survivors={s[1]} ({s[1]/max(s[0],1):.3f} of staged) "
          f"with>=1 blending pixel={s[2]} ({s[2]/max(s[1],1):.3f} of walked); pairs live={s[3]} pass={s[4]} "
          f"({s[4]/max(s[3],1):.3f}) blended={s[5]} ({s[5]/max(s[3],1):.3f}); per pixel blended={s[5]/(h*w):.1f}", flush=True)
