#!/usr/bin/env python3
"""gg_mlp_bwd at the reference's training size (1000 sampled points, 32 -> 128 -> 512), hipEvent-timed.
    python tools/mlp_bwd_bench.py [--lib variant.so] [--rows 1000] [--label text]"""
import argparse, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT]
import torch
from gaussiangrasper_amd import _lib
ap = argparse.ArgumentParser()
ap.add_argument("--lib", default=None)
ap.add_argument("--rows", type=int, default=1000)
ap.add_argument("--in-dim", type=int, default=32)
ap.add_argument("--label", default="")
a = ap.parse_args()
if a.lib:
    _lib.LIB_PATH = os.path.abspath(a.lib)
_lib.load(build_if_missing=False)
from gaussiangrasper_amd.mlp import mlp_forward
dev = "cuda:0"
g = torch.Generator().manual_seed(0)
x = torch.randn(a.rows, a.in_dim, generator=g).to(dev).requires_grad_(True)
w1 = (torch.randn(128, a.in_dim, generator=g) * 0.3).to(dev).requires_grad_(True)
b1 = torch.randn(128, generator=g).to(dev).requires_grad_(True)
w2 = (torch.randn(512, 128, generator=g) * 0.2).to(dev).requires_grad_(True)
b2 = torch.randn(512, generator=g).to(dev).requires_grad_(True)
v = torch.randn(a.rows, 512, generator=g).to(dev)
y = mlp_forward(x, w1, b1, w2, b2)
for _ in range(3):
    torch.autograd.grad(y, (x, w1, b1, w2, b2), v, retain_graph=True)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
n = 50
for _ in range(n):
    torch.autograd.grad(y, (x, w1, b1, w2, b2), v, retain_graph=True)
e1.record()
torch.cuda.synchronize()
print(json.dumps({"label": a.label, "rows": a.rows, "mlp_backward_us_per_call_incl_memsets": round(1e3 * e0.elapsed_time(e1) / n, 1)}))
