#!/usr/bin/env python3
"""Image-space main loss (L1 + SSIM, reference gaussian_splatting.py:882-885, :931) at 1600x1200: the fused HIP
pass each way against the caller's torch ops (F.conv2d separable SSIM as pytorch_msssim does it) on the same GPU.
Prints one JSON object with HBM roofline figures.  Usage: python tools/loss_bench.py [--reps 20]"""
import argparse, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "shim"), os.path.join(ROOT, "tests")]
import torch
from gaussiangrasper_amd import losses
from test_image_loss import _reference_main_loss_dev


def timed(fn, reps):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--lib", default=None, help="A/B: another build of the library")
    ap.add_argument("--height", type=int, default=1200)
    ap.add_argument("--width", type=int, default=1600)
    a = ap.parse_args()
    if a.lib:
        from gaussiangrasper_amd import _lib
        _lib.LIB_PATH = os.path.abspath(a.lib)
        _lib.load(build_if_missing=False)
    dev = torch.device("cuda:0")
    h, w = a.height, a.width
    g = torch.Generator(device="cpu").manual_seed(0)
    gt = torch.rand(h, w, 3, generator=g).to(dev)
    rgb = (gt + 0.05 * torch.randn(h, w, 3, generator=g).to(dev)).clamp(0, 1)
    valid = (torch.rand(h, w, generator=g) > 0.1).to(dev)

    def ours():
        r = rgb.detach().requires_grad_(True)
        losses.main_loss(r, gt, valid, 0.2)[0].backward()

    def ours_fwd():
        losses.main_loss(rgb, gt, valid, 0.2)

    def theirs():
        r = rgb.detach().requires_grad_(True)
        _reference_main_loss_dev(r, gt, valid, 0.2)[0].backward()

    def theirs_fwd():
        with torch.no_grad():
            _reference_main_loss_dev(rgb, gt, valid, 0.2)

    t_ours, t_ours_f = timed(ours, a.reps), timed(ours_fwd, a.reps)
    t_theirs, t_theirs_f = timed(theirs, a.reps), timed(theirs_fwd, a.reps)
    px = h * w
    bytes_fwd, bytes_bwd = px * (24 + 1 + 36), px * (36 + 24 + 1 + 12)
    out = {"image": [h, w], "reps": a.reps,
           "hip_fwd_bwd_ms": round(t_ours, 4), "hip_fwd_ms": round(t_ours_f, 4),
           "torch_fwd_bwd_ms": round(t_theirs, 4), "torch_fwd_ms": round(t_theirs_f, 4),
           "speedup_fwd_bwd": round(t_theirs / t_ours, 2),
           "roofline": {"bound": "hbm", "unit": "GB/s", "peak": 8000.0,
                        "algorithmic_bytes": {"fwd": bytes_fwd, "bwd": bytes_bwd},
                        "achieved_fwd_bwd": round((bytes_fwd + bytes_bwd) / (t_ours * 1e-3) / 1e9, 1),
                        "frac_fwd_bwd": round((bytes_fwd + bytes_bwd) / (t_ours * 1e-3) / 1e9 / 8000.0, 4),
                        "note": "times include the autograd.Function host overhead and the allocations of one call"}}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
