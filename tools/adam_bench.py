#!/usr/bin/env python3
"""Timing of the optimizer-side kernels of SURVEY 8f-3 at the bench scene size (1 M Gaussians, 118 floats
each): fused Adam over the six parameter groups in one launch, cull compaction of the 6 parameters + 12
moments in one launch, densify append.  Prints one JSON line with a roofline object per kernel
(algorithmic bytes / time against the 8 TB/s HBM peak) and torch's own time for the same work."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "shim")]
import torch
from gaussiangrasper_amd import _lib
from gaussiangrasper_amd.densify import GROUPS, append_rows, compact
from gaussiangrasper_amd.optim import FusedAdam, fused_step
from gaussiangrasper_amd.scene import make_scene

dev = "cuda:0"
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
sc = make_scene(n, config_index=3).to(dev)
params = {"means": sc.means, "scales": sc.scales, "quats": sc.quats, "opacities": sc.opacities,
          "colors_all": sc.colors_all, "feature": sc.feature}
LR = {"xyz": 1.6e-4, "color": 5e-4, "feature": 5e-4, "opacity": 0.05, "scaling": 0.005, "rotation": 0.001}


def timeit(fn, reps=20):
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


floats = sum(p.numel() for p in params.values())
res = {"num_gaussians": n, "floats_per_gaussian": floats // n}
mine = {k: torch.nn.Parameter(v.clone()) for k, v in params.items()}
ref = {k: torch.nn.Parameter(v.clone()) for k, v in params.items()}
for d in (mine, ref):
    for p in d.values():
        p.grad = torch.randn_like(p) * 1e-3
o_mine = [FusedAdam([mine[a]], lr=LR[g], eps=1e-15) for g, a in GROUPS.items()]
o_ref = [torch.optim.Adam([ref[a]], lr=LR[g], eps=1e-15) for g, a in GROUPS.items()]
t = timeit(lambda: fused_step(o_mine))
t_ref = timeit(lambda: [o.step() for o in o_ref])
bytes_adam = floats * 28
res["adam"] = {"ms": t, "torch_six_optimizers_ms": t_ref, "algorithmic_bytes": bytes_adam,
               "roofline": {"bound": "hbm", "achieved": bytes_adam / t / 1e6, "peak": 8000.0, "unit": "GB/s",
                            "frac": bytes_adam / t / 1e6 / 8000.0}}
# cull: 18 arrays, 30 % deleted
arrays = []
for p in params.values():
    arrays += [p, torch.randn_like(p), torch.rand_like(p)]
mask = torch.rand(n, device=dev) < 0.3
t = timeit(lambda: compact(arrays, mask), reps=10)
t_ref = timeit(lambda: [a[~mask] for a in arrays], reps=10)
kept = int((~mask).sum())
bytes_c = 3 * floats * 4 + 3 * kept * (floats // n) * 4 + n
res["compact_18_arrays"] = {"ms": t, "torch_boolean_indexing_ms": t_ref, "algorithmic_bytes": bytes_c,
                            "roofline": {"bound": "hbm", "achieved": bytes_c / t / 1e6, "peak": 8000.0,
                                         "unit": "GB/s", "frac": bytes_c / t / 1e6 / 8000.0},
                            "note": "includes the host read-back of the kept count and the output allocations"}
# densify: 10 % split (2 samples), 15 % dup
split = torch.rand(n, device=dev) < 0.10
dup = (torch.rand(n, device=dev) < 0.15) & ~split
kinds = {"means": _lib.ROWS_MEANS, "scales": _lib.ROWS_SCALES}
arr2 = []
for k, p in params.items():
    arr2 += [(p, kinds.get(k, _lib.ROWS_COPY)), (torch.randn_like(p), _lib.ROWS_ZERO_NEW), (torch.rand_like(p), _lib.ROWS_ZERO_NEW)]
z = torch.randn(2 * int(split.sum()), 3, device=dev)
t = timeit(lambda: append_rows(arr2, split, dup, 2, z, sc.means, sc.scales, sc.quats), reps=10)
new_rows = n + 2 * int(split.sum()) + int(dup.sum())
bytes_d = 3 * floats * 4 + 3 * new_rows * (floats // n) * 4
res["densify_18_arrays"] = {"ms": t, "algorithmic_bytes": bytes_d,
                            "roofline": {"bound": "hbm", "achieved": bytes_d / t / 1e6, "peak": 8000.0,
                                         "unit": "GB/s", "frac": bytes_d / t / 1e6 / 8000.0},
                            "note": "includes two rank scans with their read-backs and the output allocations"}
print(json.dumps(res))
