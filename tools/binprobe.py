#!/usr/bin/env python3
"""Projection + binning only (no blend), three times, on the bench view — for rocprofv3 kernel traces of binning variants:
    rocprofv3 --kernel-trace --stats --output-format csv -d out -- python3 tools/binprobe.py [path/to/variant.so]"""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "shim")]
import torch  # noqa: E402
from gaussiangrasper_amd import _lib, ops  # noqa: E402
from gaussiangrasper_amd.camera import ring_cameras  # noqa: E402
from gaussiangrasper_amd.scene import make_scene  # noqa: E402
args = [a for a in sys.argv[1:] if not a.startswith("--")]
if args:
    _lib.LIB_PATH = os.path.abspath(args[0])
dev = "cuda:0"
h, w, n = 1200, 1600, 1_000_000
if "--config5" in sys.argv:            # BASELINE config 5: 5 M Gaussians at 1920x1080
    h, w, n = 1080, 1920, 5_000_000
sc = make_scene(n, config_index=5 if "--config5" in sys.argv else 3).to(dev)
v = ring_cameras(8, h, w, device=dev)[0]
with torch.no_grad():
    xys, depths, radii, conics, nth, _ = ops.ProjectGaussians.apply(
        sc.means, sc.scales.exp(), 1, sc.quats, v.viewmat[:3], v.projmat, v.fx, v.fy, v.cx, v.cy, h, w, v.tile_bounds)
    for _ in range(3):
        ops.clear_bin_cache()
        b = ops.bin_and_sort_gaussians(xys, depths, radii, nth, h, w)
    torch.cuda.synchronize()
print("num_intersects", b.num_intersects)
