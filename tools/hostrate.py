#!/usr/bin/env python3
"""How long the host takes to enqueue a view against how long the GPU takes to run it (bench workload, plugin route),
for the sequential step and for the two-stream pipelined step (dist.train_step_pipelined).  r02: 2.05 / 2.10 ms
sequential, 1.97 / 2.04 ms pipelined — the host is held to the GPU's pace by the per-view count check."""
import sys, time, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'shim')]
import torch
from gaussiangrasper_amd import ops
from gaussiangrasper_amd.camera import ring_cameras
from gaussiangrasper_amd.scene import make_scene
from gaussiangrasper_amd.dist import GradBucket, train_step, train_step_pipelined
from gaussiangrasper_amd.pipeline import backward_view, render_view, seeded_cotangents
dev = torch.device("cuda", 0)
h, w = 1200, 1600
scene = make_scene(1_000_000, feature_dim=32, config_index=3).to(dev)
for p in scene.params(): p.requires_grad_(True)
views = ring_cameras(8, h, w, device=dev)
bucket = GradBucket(scene.params()); bucket.enable_direct(ops, defer_sh=True)
probe = render_view(scene, views[0], ops, fused=True)
cot = seeded_cotangents(probe, seed=1234); del probe
ops.clear_bin_cache()
render = lambda v: render_view(scene, views[v], ops, fused=True)
backward = lambda out: backward_view(out, cot)
def fn(v): backward(render(v))
streams = [torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)]
for mode in ("seq", "pipe", "seq", "pipe"):
    for _ in range(2):
        (train_step(fn, bucket, range(8)) if mode == "seq" else train_step_pipelined(render, backward, bucket, range(8), streams))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        (train_step(fn, bucket, range(8)) if mode == "seq" else train_step_pipelined(render, backward, bucket, range(8), streams))
    t_host = time.perf_counter() - t0
    torch.cuda.synchronize()
    t_all = time.perf_counter() - t0
    print(f"{mode}: host enqueue {1e3 * t_host / 40:.3f} ms per view, with GPU {1e3 * t_all / 40:.3f} ms per view", flush=True)
