#!/usr/bin/env python3
"""In-kernel cycle stamps of the wide backward kernels on the bench view (diagnostic build -DGG_STAMPS: read its
SHARES, never its run time).  Per wave and per batch: cycles spent in staging, the colour-row load wait, the D
product, the walk, the flush MFMAs, the colour atomics, the second array's flush, queue compaction."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "shim")]
import torch
from gaussiangrasper_amd import _lib, build as gg_build, ops
from gaussiangrasper_amd.camera import ring_cameras
from gaussiangrasper_amd.scene import make_scene

out = os.path.join(os.path.dirname(gg_build.OUT), "libgg_raster_stamps.so")
if os.environ.get("STAMPS_LIB"):      # a prebuilt diagnostic variant (e.g. -DGG_EPI_SKIP=1)
    out = os.path.abspath(os.environ["STAMPS_LIB"])
else:
    gg_build.build(force=True, extra_flags=("-DGG_STAMPS",), out=out)
_lib.LIB_PATH = out
lib = _lib.load()
# a -DGG_EPI_SKIP=1 forward stores no final_T / final_idx: no backward may run on its outputs (the build says so)
lib.gg_debug_epi_skip.restype = ctypes.c_int
FWD_ONLY = lib.gg_debug_epi_skip() == 1
dev = "cuda:0"
h, w, n = 1200, 1600, 1_000_000
sc = make_scene(n, config_index=3).to(dev)
v = ring_cameras(8, h, w, device=dev)[0]
xys, depths, radii, conics, nth, _ = ops.ProjectGaussians.apply(
    sc.means, sc.scales.exp(), 1, sc.quats, v.viewmat[:3], v.projmat, v.fx, v.fy, v.cx, v.cy, h, w, v.tile_bounds)
opac = torch.sigmoid(sc.opacities)
names = ["prologue", "staging", "colour-row wait", "D product", "walk", "flush MFMAs", "colour atomics",
         "second-array flush", "queue compaction", "wave lifetime"]


def report(label):
    torch.cuda.synchronize()
    buf = (ctypes.c_ulonglong * 16)()
    lib.gg_debug_stamps(buf, 1)
    s = list(buf)
    waves, batches, life = max(s[11], 1), max(s[10], 1), max(s[9], 1)
    print(f"{label}: waves {waves}, batches {batches} ({batches / waves:.2f} per wave), "
          f"cycles per wave {life / waves:.0f}, per batch {life / batches:.0f}")
    for i in range(9):
        print(f"    {names[i]:20s} {s[i] / life:6.3f} of the wave's lifetime, {s[i] / batches:8.0f} cycles per batch")
    print(f"    {'(unaccounted)':20s} {1 - sum(s[:9]) / life:6.3f}", flush=True)


feat = sc.feature.detach().requires_grad_(True)
x = xys.detach().requires_grad_(True)
vo = torch.randn(h, w, 32, device=dev)
for rep in range(2):
    torch.cuda.synchronize()
    lib.gg_debug_stamps(None, 1)
    out_ = ops.NDRasterizeGaussians.apply(x, depths, radii, conics.detach(), nth, feat, opac.detach(), h, w,
                                          torch.zeros(32, device=dev))
    if rep == 1:
        report("32-channel FORWARD (phases: 0 prologue, 1 staging, 4 walk, 7 epilogue; a batch = a chunk of 64 list entries)")
    lib.gg_debug_stamps(None, 1)
    if not FWD_ONLY:
        out_.backward(vo)
if not FWD_ONLY:
    report("32-channel backward")
tail = torch.rand(n, 7, device=dev).requires_grad_(True)
vos = [torch.randn(h, w, 32, device=dev), torch.randn(h, w, 7, device=dev)]
for rep in range(2):
    torch.cuda.synchronize()
    lib.gg_debug_stamps(None, 1)
    imgs = ops.rasterize_segments(x, depths, radii, conics.detach(), nth, opac.detach(), h, w,
                                  [(feat, torch.zeros(32, device=dev)), (tail, torch.zeros(7, device=dev))])
    if rep == 1:
        report("pair FORWARD (32 + 7)")
    lib.gg_debug_stamps(None, 1)
    if not FWD_ONLY:
        torch.autograd.backward(imgs, vos)
if not FWD_ONLY:
    report("pair backward (32 + 7)")
