#!/usr/bin/env python3
"""bench.py — views/s of the splatting hot path on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W
    (N>1: launched by `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...`)

Workload (BASELINE config 4, per GPU): 1 M synthetic Gaussians (SURVEY §8d generator), 1600x1200,
SH degree 4, 32-ch feature; per view the reference's operator sequence
(nerfstudio/models/gaussian_splatting.py:699-784): ProjectGaussians -> SphericalHarmonics ->
RasterizeGaussians(rgb) -> NDRasterizeGaussians(32-ch feature) -> RasterizeGaussians(depth) ->
RasterizeGaussians(normal), then ONE backward with dense N(0,1) cotangents on all four images,
through the gsplat-compatible operators of gaussiangrasper_amd (the drop-in path).

A step = `--views-per-step` (8) views per rank, forward+backward with local gradient
accumulation, then one RCCL all-reduce of the 472 MB parameter-gradient buffer when N>1
(SURVEY §8e).  Weak scaling: per-GPU work is fixed, N=8 is exactly config 4 (64 views).
Inputs are resident in HBM before the timed region.  Rank 0 prints ONE JSON line.

Extra objects in the line:
  roofline     — the dominant kernel (blend_bwd_kernel<32>, the 32-ch feature backward):
                 achieved = algorithmic bytes per launch / average launch duration, the duration
                 measured live with hipEvents bracketing exactly that kernel over the timed
                 region (gg_prof_* in the C ABI).  Byte formula: DESIGN.md §5.
  cpu_baseline — the CPU oracle ("port", C + OpenMP, all host cores) timed on ONE view of the same
                 workload (same operator sequence, fwd+bwd), rank 0, N=1 only.
"""
from __future__ import annotations

import argparse
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "shim")):
    if p not in sys.path:
        sys.path.insert(0, p)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--points", type=int, default=1_000_000)
    ap.add_argument("--height", type=int, default=1200)
    ap.add_argument("--width", type=int, default=1600)
    ap.add_argument("--feature-dim", type=int, default=32)
    ap.add_argument("--views-per-step", type=int, default=8)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-prof", action="store_true", help="do not bracket kernels with hipEvents")
    ap.add_argument("--no-fused", action="store_true", help="skip the secondary fused-path measurement")
    return ap.parse_args()


def algorithmic_bytes_blend_bwd(P: int, I: int, n_vis: int, C: int) -> int:
    """Bytes one blend_bwd launch with C channels must move (SURVEY §8d terms 2-4
    restricted to this kernel): v_out + final_T + final_idx per pixel, the sorted id list, and per
    visible Gaussian its 2-D record (xy 8, conic 12, opacity 4), its C colours, and the gradients
    written back (C colours + xy 2 + conic 3 + opacity 1)."""
    return P * (4 * C + 8) + 4 * I + n_vis * (24 + 4 * C) + n_vis * (4 * C + 24)


def cpu_baseline(args, scene, view):
    """One view fwd+bwd through the oracle-backed operators on the host cores."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_ops
    from oracle import oracle as O
    from gaussiangrasper_amd.pipeline import backward_view, render_view, seeded_cotangents
    O.build()
    cores = os.cpu_count() or 1
    O.set_num_threads(cores)
    torch.set_num_threads(cores)
    sc = scene.to("cpu")
    for p in sc.params():
        p.requires_grad_(True)
    t0 = time.perf_counter()
    out = render_view(sc, view, oracle_ops)
    backward_view(out, seeded_cotangents(out, seed=0))
    dt = time.perf_counter() - t0
    return {"value": 1.0 / dt, "unit": "views/s", "cores": O.num_threads(), "kind": "port",
            "sample": f"1 view fwd+bwd of the same workload ({args.points} Gaussians, "
                      f"{args.width}x{args.height}, rgb+feature{args.feature_dim}+depth+normal), "
                      f"oracle/gg_oracle.c with OpenMP, {dt:.1f} s wall"}


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback in the product path)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)
    assert world == args.gpus or world == 1, f"WORLD_SIZE {world} != --gpus {args.gpus}"

    from gaussiangrasper_amd import _lib, ops
    from gaussiangrasper_amd.camera import ring_cameras
    from gaussiangrasper_amd.dist import GradBucket, shard_views, train_step
    from gaussiangrasper_amd.pipeline import backward_view, render_view, seeded_cotangents
    from gaussiangrasper_amd.scene import make_scene

    lib = _lib.load()
    scene_cpu = make_scene(args.points, feature_dim=args.feature_dim, config_index=3)
    scene = scene_cpu.to(dev)
    for p in scene.params():
        p.requires_grad_(True)
    total_views = args.views_per_step * world
    views = ring_cameras(total_views, args.height, args.width, device=dev)
    my_views = shard_views(total_views, rank, world)
    bucket = GradBucket(scene.params())
    # cotangents resident in HBM, one set reused for every view (dense N(0,1), seeded)
    probe = render_view(scene, views[my_views[0]], ops)
    cot = seeded_cotangents(probe, seed=1234)
    n_vis = int((probe["radii"] > 0).sum())
    n_isect = int(probe["num_tiles_hit"].long().sum())
    del probe
    ops.clear_bin_cache()

    def render_and_backward(v):
        out = render_view(scene, views[v], ops)
        backward_view(out, cot)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        train_step(render_and_backward, bucket, my_views)
    barrier()
    if not args.no_prof:
        lib.gg_prof_reset()
        lib.gg_prof_enable(1)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        train_step(render_and_backward, bucket, my_views)
    barrier()
    elapsed = time.perf_counter() - t0
    lib.gg_prof_enable(0)
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # secondary measurement (not `value`): same views through the fused single-call path (§8f-1)
    fused_vps = None
    if not args.no_fused:
        def render_and_backward_fused(v):
            out = render_view(scene, views[v], ops, fused=True)
            backward_view(out, cot)
        train_step(render_and_backward_fused, bucket, my_views)   # warm-up
        barrier()
        tf0 = time.perf_counter()
        for _ in range(args.steps):
            train_step(render_and_backward_fused, bucket, my_views)
        barrier()
        tf = time.perf_counter() - tf0
        if world > 1:
            t = torch.tensor([tf], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            tf = float(t.item())
        fused_vps = total_views * args.steps / tf

    views_done = total_views * args.steps
    result = {
        "metric": "rendered views/sec (fwd+bwd) at 1M Gaussians, 1600x1200, 32-ch feature",
        "value": views_done / elapsed, "unit": "views/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": "BASELINE config 4 per-GPU share: %d Gaussians, %dx%d, SH deg 4 rgb + "
                               "%d-ch feature + depth + normal via the reference's 4 rasterize calls, "
                               "fwd+bwd, %d views/step/GPU, 1 grad all-reduce/step"
                               % (args.points, args.width, args.height, args.feature_dim,
                                  args.views_per_step),
                   "num_gaussians": args.points, "image": [args.height, args.width],
                   "feature_dim": args.feature_dim, "views_per_step_per_gpu": args.views_per_step,
                   "n_visible": n_vis, "num_intersects": n_isect,
                   "parallelism": f"view-parallel x{world}, replicated Gaussians",
                   "grad_allreduce_bytes": bucket.nbytes},
    }

    if rank == 0:
        # per-kernel averages over the timed region (hipEvents inside the library)
        kernels = {}
        if not args.no_prof:
            for kid in range(32):
                n, ms = ctypes.c_int(0), ctypes.c_double(0.0)
                lib.gg_prof_get(kid, ctypes.byref(n), ctypes.byref(ms))
                if n.value:
                    kernels[lib.gg_prof_name(kid).decode()] = {
                        "launches": n.value, "avg_ms": ms.value / n.value, "total_ms": ms.value}
            lib.gg_prof_reset()
        dom = "blend_bwd_kernel<32>"
        roofline = None
        if dom in kernels:
            P = args.height * args.width
            b_alg = algorithmic_bytes_blend_bwd(P, n_isect, n_vis, 32)
            achieved = b_alg / (kernels[dom]["avg_ms"] * 1e-3) / 1e9
            traffic = valu_busy = None
            tfile = os.path.join(ROOT, "profiles", "traffic.json")   # from separate --pmc runs
            if os.path.exists(tfile):
                try:
                    pmc = json.load(open(tfile))
                    traffic = pmc.get(dom)
                    valu_busy = pmc.get("valu_busy", {}).get(dom)
                except Exception:  # noqa: BLE001
                    traffic = valu_busy = None
            roofline = {"bound": "hbm", "kernel": dom, "achieved": achieved, "peak": HBM_PEAK_GBS,
                        "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                        "algorithmic_bytes_per_launch": b_alg,
                        # the kernel is VALU-issue bound: share of VALU issue slots used (PMC pass)
                        "valu_busy": valu_busy,
                        "avg_launch_ms": kernels[dom]["avg_ms"], "launches": kernels[dom]["launches"]}
        result["roofline"] = roofline
        result["kernels"] = {k: round(v["avg_ms"], 4) for k, v in sorted(kernels.items())}
        tot = sum(v["total_ms"] for v in kernels.values())
        result["kernel_time_fraction_of_wall"] = tot / (1e3 * elapsed) if kernels else None
        result["fused_single_call_path"] = None if fused_vps is None else {
            "value": fused_vps, "unit": "views/s",
            "note": "same views and gradients through ONE NDRasterize call on feature|rgb|depth|normal "
                    "(SURVEY 8f-1, pipeline.rasterize_activated_fused); NOT the headline: the headline "
                    "is the reference's unchanged 4-call sequence"}
        result["cpu_baseline"] = None
        if world == 1 and not args.no_cpu_baseline:
            try:
                result["cpu_baseline"] = cpu_baseline(args, scene_cpu, ring_cameras(
                    total_views, args.height, args.width)[0])
            except Exception as exc:  # noqa: BLE001
                result["cpu_baseline"] = {"error": repr(exc)}
        print(json.dumps(result), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
