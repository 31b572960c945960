#!/usr/bin/env python3
"""Where one training iteration's host time goes (tools/train_step_bench.py's fused iteration under cProfile, with a
device synchronisation after each of render / get_loss_dict / backward so that the wall clock of each is its own)."""
import cProfile
import io
import os
import pstats
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "shim"), os.path.join(ROOT, "tests")]
import torch  # noqa: E402

from gaussiangrasper_amd.camera import ring_cameras  # noqa: E402
from gaussiangrasper_amd.scene import make_scene  # noqa: E402
from gaussiangrasper_amd.plugin import make_fused_model_class  # noqa: E402
from gaussiangrasper_amd.stub import StubCameras, StubGaussianSplattingModel  # noqa: E402

dev = torch.device("cuda:0")
h, w, n = 1200, 1600, 1_000_000
scene = make_scene(n, config_index=3).to(dev)
views = ring_cameras(8, h, w, device=dev)
torch.manual_seed(3)
model = make_fused_model_class(StubGaussianSplattingModel, fused_training=True)(scene).to(dev).train()
cams = [StubCameras.from_view(v, device=dev, cam_idx=i) for i, v in enumerate(views)]
g = torch.Generator(device="cpu").manual_seed(7)
hs, ws = h // 2, w // 2
yy, xx = torch.meshgrid(torch.arange(hs), torch.arange(ws), indexing="ij")
depth = torch.rand(hs, ws, 1, generator=g) * 5 + 0.5
batch = {"image": torch.rand(h, w, 3, generator=g), "normal": torch.randn(hs, ws, 3, generator=g), "depth": depth,
         "sam_mask": ((yy * 2) // hs + 2 * ((xx * 2) // ws)).float() - 1.0,
         "valid_mask": torch.rand(hs, ws, generator=g) > 0.05, "feature": torch.randn(h // 8, w // 8, 512, generator=g)}
batch = {k: v.to(dev) for k, v in batch.items()}
sync = torch.cuda.synchronize
T = {"render": 0.0, "loss": 0.0, "backward": 0.0}


def one(k):
    sync(); t0 = time.perf_counter()
    out = model(cams[k])
    out["rgb"]
    sync(); t1 = time.perf_counter()
    ld = model.get_loss_dict(out, batch)
    sync(); t2 = time.perf_counter()
    sum(ld.values()).backward()
    sync(); t3 = time.perf_counter()
    T["render"] += t1 - t0; T["loss"] += t2 - t1; T["backward"] += t3 - t2


for k in range(2):
    one(k)
for k in T:
    T[k] = 0.0
pr = cProfile.Profile()
pr.enable()
for k in range(4):
    one(k)
pr.disable()
print({k: round(1e3 * v / 4, 2) for k, v in T.items()}, "ms per view")
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(45)
print(s.getvalue()[:9000])
