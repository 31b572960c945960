#!/usr/bin/env python3
"""bench.py — views/s of the splatting hot path on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W

N > 1: one process per GPU over RCCL.  Either the driver starts the ranks
(`python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...`: RANK / LOCAL_RANK /
WORLD_SIZE / MASTER_* in the environment), or — plain `python bench.py --gpus N` — this file starts
them itself: the parent spawns N fresh children BEFORE it touches the GPU (it never initialises HIP
and nothing is re-exec'd), waits for them and exits with their status.  WORLD_SIZE must equal
--gpus; there is no silent single-GPU fallback.

Workload (BASELINE config 4, per GPU): 1 M synthetic Gaussians (SURVEY §8d generator), 1600x1200,
SH degree 4, 32-ch feature; per view the reference's operator sequence
(nerfstudio/models/gaussian_splatting.py:699-784): ProjectGaussians -> SphericalHarmonics ->
RasterizeGaussians(rgb) -> NDRasterizeGaussians(32-ch feature) -> RasterizeGaussians(depth) ->
RasterizeGaussians(normal), then ONE backward with dense N(0,1) cotangents on all four images,
through the gsplat-compatible operators of gaussiangrasper_amd (the drop-in path).

A step = `--views-per-step` (8) views per rank, forward+backward with local gradient
accumulation, then the RCCL reduction of the 472 MB parameter-gradient buffer when N>1 (per
parameter, overlapped with the last view's backward).  Weak scaling: per-GPU work is
fixed, N=8 is exactly config 4 (64 views).  Inputs are resident in HBM before the timed region.
Rank 0 prints ONE JSON line.

`--config 5`: BASELINE config 5's per-GPU share instead — render.sh's body
(nerfstudio/pipelines/base_pipeline.py:401-408) on 5 M Gaussians, 128-dim feature, 1920x1080,
render-only: project, SH, the four rasterize forwards and the fea_up MLP on every pixel; no
collective.  Its line carries the same keys (metric "rendered views/sec (render-only) ...").

The timed region runs with the in-library kernel timers OFF; one extra profiled step afterwards
(hipEvent pairs from a pool, recorded on the launch stream) gives the per-kernel averages.

Extra objects in the line:
  roofline     — the dominant kernel BY TOTAL TIME over the profiled step: achieved = algorithmic
                 bytes per launch / average launch duration (byte formulas: DESIGN.md §5), peak 8 TB/s;
                 `kernels` holds the same for every blend kernel; `whole_view` prices SURVEY §8d's
                 whole-view bytes (recomputed from the measured N_vis and I) against the wall time
                 per view; `valu` prices the kernel against the VALU issue peak (instruction counts
                 and weights from the PMC pass under profiles/).
  cpu_baseline — the CPU oracle ("port", C + OpenMP, all host cores) timed on ONE view of the same
                 workload (same operator sequence, fwd+bwd), rank 0, N=1 only.

Test hooks (used only by tests/): `--backend gloo --device cpu --ops <module>` runs the same
launcher / sharding / reduction code on CPU tensors with a non-product operator module (the line is
then marked "selftest"); `--dump-grads PATH` saves rank 0's reduced gradient bucket.
"""
from __future__ import annotations

import argparse
import ctypes
import importlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "shim")):
    if p not in sys.path:
        sys.path.insert(0, p)

HBM_PEAK_GBS = 8000.0       # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
PROFILED_STEPS = 2          # steps behind the timed region that run with the kernel timers on (one stream, no overlap)
SIMDS = 256 * 4             # 256 CUs x 4 SIMD-32
CLOCK_HZ = 2.4e9            # max clock; a plain wave64 VALU instruction issues over 2 cycles
PRODUCT_OPS = "gaussiangrasper_amd.ops"


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", type=int, default=4, choices=(4, 5))
    ap.add_argument("--points", type=int, default=None)
    ap.add_argument("--height", type=int, default=None)
    ap.add_argument("--width", type=int, default=None)
    ap.add_argument("--feature-dim", type=int, default=None)
    ap.add_argument("--views-per-step", type=int, default=8)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-config5", action="store_true",
                    help="skip the short BASELINE-config-5 (render-only) object of the default one-GPU line")
    ap.add_argument("--no-prof", action="store_true", help="skip the profiled step (no per-kernel times)")
    ap.add_argument("--torch-profile", default=None, metavar="PATH",
                    help="after the timed steps, one more step under torch.profiler (shapes recorded); the per-operator "
                         "table goes to PATH — which torch operators launch what, beside the library's own kernels")
    ap.add_argument("--route", default="plugin", choices=("plugin", "shim"),
                    help="headline route: the nerfstudio plugin's fused operator (default) or the shim's "
                         "four separate rasterize calls; the other one is measured as well")
    ap.add_argument("--no-secondary", "--no-fused", dest="no_secondary", action="store_true",
                    help="skip the measurement of the other route")
    ap.add_argument("--no-overlap", action="store_true", help="one all-reduce after the last view")
    ap.add_argument("--reduce", default="allreduce", choices=("allreduce", "rs_ag"),
                    help="gradient-to-Gaussian reduction: all-reduce of the gradient bucket (default; the timed step "
                         "ends with reduced gradients, as the reference's DDP step would) or rs_ag = reduce-scatter, "
                         "fused Adam on this rank's shard, all-gather of the parameters (dist.ShardedAdamStep; the "
                         "timed step then INCLUDES the optimizer step)")
    ap.add_argument("--keep-gc", action="store_true",
                    help="do not gc.freeze() after the warm-up (see freeze_gc)")
    ap.add_argument("--pipe-priority", default="0,0",
                    help="measurement: HIP stream priorities of the two view-pipeline streams (-1 = high)")
    ap.add_argument("--no-view-pipeline", action="store_true",
                    help="the views of a step strictly one after the other on one stream (default: backward of "
                         "view k beside forward of view k + 1 on two streams)")
    ap.add_argument("--no-direct", action="store_true",
                    help="autograd accumulates every parameter gradient (no in-kernel accumulation "
                         "into the gradient bucket for the SH and feature parameters)")
    ap.add_argument("--deterministic", action="store_true",
                    help="bit-reproducible blend gradients (ops.set_deterministic_backward): a measurement of "
                         "that mode's cost, never the headline")
    ap.add_argument("--backend", default="nccl", choices=("nccl", "gloo"))
    ap.add_argument("--device", default="cuda", choices=("cuda", "cpu"))
    ap.add_argument("--ops", default=PRODUCT_OPS, help="operator module (tests only)")
    ap.add_argument("--dump-grads", default=None)
    ap.add_argument("--share-gpu", action="store_true",
                    help="test hook: every rank uses cuda:0 (a one-GPU box rehearsing N ranks; gloo backend)")
    a = ap.parse_args(argv)
    if a.keep_gc:
        os.environ["GG_BENCH_KEEP_GC"] = "1"
    d4 = dict(points=1_000_000, height=1200, width=1600, feature_dim=32)
    d5 = dict(points=5_000_000, height=1080, width=1920, feature_dim=128)
    for k, v in (d4 if a.config == 4 else d5).items():
        if getattr(a, k) is None:
            setattr(a, k, v)
    return a


# ------------------------------------------------------------------------------------------------
# algorithmic bytes (DESIGN.md §5 / SURVEY §8d)
# ------------------------------------------------------------------------------------------------
def algorithmic_bytes_blend_bwd(P: int, I: int, n_vis: int, C: int) -> int:
    """One blend_bwd launch with C channels: v_out + final_T + final_idx per pixel, the sorted id
    list, per visible Gaussian its 2-D record (xy 8, conic 12, opacity 4) + C colours read and the
    gradients written back (C colours + xy 2 + conic 3 + opacity 1)."""
    return P * (4 * C + 8) + 4 * I + n_vis * (24 + 4 * C) + n_vis * (4 * C + 24)


def algorithmic_bytes_blend_fwd(P: int, I: int, n_vis: int, C: int) -> int:
    """One blend_fwd launch with C channels: out + final_T + final_idx written per pixel, the sorted
    id list and per visible Gaussian its 2-D record + C colours read."""
    return P * (4 * C + 8) + 4 * I + n_vis * (24 + 4 * C)


def algorithmic_bytes_whole_view(N: int, n_vis: int, I: int, P: int, K: int, D: int,
                                 render_only: bool = False) -> int:
    """SURVEY §8d B_alg per view: (1) parameters read (+ gradients written), (2) 56-byte 2-D
    intermediates per visible Gaussian, (3) intersection keys / sort / list reads, (4) pixels;
    depth counted once (C_alg = 3 + 1 + 3 + D)."""
    c_alg = 7 + D
    par = N * (44 + 12 * K + 4 * D)
    if render_only:
        return par + 2 * 56 * n_vis + 40 * I + P * 4 * c_alg
    return 2 * par + 5 * 56 * n_vis + 44 * I + 2 * P * (4 * c_alg + 8)


# ------------------------------------------------------------------------------------------------
def cpu_baseline(args, scene, view):
    """One view fwd+bwd through the oracle-backed operators on the host cores."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch
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


def cpu_baseline_render(args, scene, view, fea_up):
    """Config 5 on the host cores, BOUNDED: one full-size view forward on the CPU oracle is minutes (30.7 M list
    entries x 135 channels + a 2 M-pixel 128 -> 128 -> 512 MLP), so the sample is
      (a) projection + SH of ALL Gaussians for the view (full size), and
      (b) binning, the four rasterize forwards and fea_up on a WINDOW of the same view — the same camera and
          intrinsics with the principal point shifted so that a (W / 4) x (H / 4) rectangle around the image centre
          is the whole image: the same Gaussians per pixel as the full view, 1 / 16 of the pixels;
    value = 1 / (t_a + 16 t_b).  Both on the oracle (oracle/gg_oracle.c, OpenMP on every host core) through the
    shim's operator sequence, as cpu_baseline() does for config 4."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch
    import oracle_ops
    from oracle import oracle as O
    from gaussiangrasper_amd.camera import view_from_c2w
    from gaussiangrasper_amd.pipeline import activate, render_view
    O.build()
    cores = os.cpu_count() or 1
    O.set_num_threads(cores)
    torch.set_num_threads(cores)
    sc = scene.to("cpu")
    w4, h4 = max(16, view.width // 4 // 16 * 16), max(16, view.height // 4 // 16 * 16)
    x0, y0 = (view.width - w4) // 2, (view.height - h4) // 2
    c2w = torch.eye(4)
    wv = view.viewmat.detach().cpu()
    c2w[:3, :3] = wv[:3, :3].T @ torch.diag(torch.tensor([1.0, -1.0, -1.0]))
    c2w[:3, 3] = view.cam_pos.detach().cpu()
    window = view_from_c2w(c2w, view.fx, view.fy, view.cx - x0, view.cy - y0, h4, w4)
    full = view_from_c2w(c2w, view.fx, view.fy, view.cx, view.cy, view.height, view.width)
    with torch.no_grad():
        t0 = time.perf_counter()
        act = activate(sc, full, oracle_ops.quat_to_rotmat)
        oracle_ops.ProjectGaussians.apply(act["means"], act["scales"], 1, act["quats"], full.viewmat[:3, :],
                                          full.projmat, full.fx, full.fy, full.cx, full.cy, full.height, full.width,
                                          full.tile_bounds)
        oracle_ops.SphericalHarmonics.apply(4, act["viewdirs"], act["sh"])
        t_a = time.perf_counter() - t0
        t0 = time.perf_counter()
        out = render_view(sc, window, oracle_ops)
        feat = out["feature"].reshape(-1, out["feature"].shape[-1]).numpy()
        l0, l2 = fea_up.layers[0], fea_up.layers[2]
        O.mlp_fwd(feat, l0.weight.detach().cpu().numpy(), l0.bias.detach().cpu().numpy(),
                  l2.weight.detach().cpu().numpy(), l2.bias.detach().cpu().numpy())
        t_b = time.perf_counter() - t0
        # (the window's own projection + SH are inside t_b as well: counted once too often, 1 / 16 of t_a)
    scale = (view.width * view.height) / float(w4 * h4)
    est = t_a + scale * t_b
    return {"value": 1.0 / est, "unit": "views/s", "cores": O.num_threads(), "kind": "port",
            "sample": f"bounded: projection + SH of all {args.points} Gaussians at full size ({t_a:.1f} s) + binning, "
                      f"four rasterize forwards (rgb, {args.feature_dim}-ch feature, depth, normal) and fea_up on a "
                      f"{w4}x{h4} window of the {view.width}x{view.height} view ({t_b:.1f} s, x{scale:.1f} pixels); "
                      f"extrapolated view time {est:.1f} s; oracle/gg_oracle.c with OpenMP"}


def parse_rccl_log(path):
    """what RCCL's INFO log (NCCL_DEBUG_FILE of rank 0) says about algorithms / protocols / channels: a few distinct
    lines, or None (no log, gloo, or a build that prints nothing of the kind)"""
    if not path or not os.path.exists(path):
        return None
    keep, seen = [], set()
    try:
        with open(path, errors="replace") as f:
            for line in f:
                low = line.lower()
                if any(k in low for k in ("algo", "proto", "channel", "ring", "tree", "xgmi", "p2p")) and "via" not in low:
                    txt = line.split("NCCL INFO", 1)[-1].strip()[:160]
                    key = "".join(c for c in txt if not c.isdigit())
                    if key not in seen:
                        seen.add(key)
                        keep.append(txt)
                if len(keep) >= 24:
                    break
    except OSError:
        return None
    return keep or None


def freeze_gc():
    """Python's cyclic collector walks every tracked object of the process on a full collection — ~60 ms with torch
    loaded, once per ~50-70 views at ~1 000 container allocations per view — and the GPU drains meanwhile (the host is at
    most one view ahead of it: the intersection count is read back per view).  Measured on `--config 5`: 157 views/s
    with the collector as it comes, 203 with it off.  After the warm-up everything alive is moved to the permanent
    generation (`gc.freeze()`): later collections only walk what the timed steps allocate.  `--keep-gc` leaves the
    collector untouched."""
    import gc
    if os.environ.get("GG_BENCH_KEEP_GC") == "1":
        return
    gc.collect()
    gc.freeze()


def read_kernel_times(lib):
    kernels = {}
    for kid in range(40):
        n, ms = ctypes.c_int(0), ctypes.c_double(0.0)
        lib.gg_prof_get(kid, ctypes.byref(n), ctypes.byref(ms))
        if n.value:
            kernels[lib.gg_prof_name(kid).decode()] = {
                "launches": n.value, "avg_ms": ms.value / n.value, "total_ms": ms.value}
    lib.gg_prof_reset()
    return kernels


def load_pmc():
    path = os.path.join(ROOT, "profiles", "pmc.json")
    if not os.path.exists(path):
        path = os.path.join(ROOT, "profiles", "traffic.json")
    try:
        return json.load(open(path)), os.path.relpath(path, ROOT)
    except Exception:  # noqa: BLE001
        return {}, None


# cycles one MFMA holds the matrix pipe (to turn busy cycles into an instruction count): the exact-order forward kernels
# use v_mfma_f32_32x32x2_f32 (64), the batched pair forward (r04, the default) v_mfma_f32_16x16x32_f16 (16); the 16-slot
# backward kernels v_mfma_f32_16x16x32_f16 (16) for D and the colour flush
# and, in the pair build, v_mfma_f32_16x16x4_f32 (32) for the second array: 32 x 16 + 24 x 32 cycles per batch of 56
MFMA_CYCLES_F32 = {"blend_fwd_pair_kernel<40>": 16.0, "blend_fwd_kernel<32>": 64.0,
                   "blend_bwd_pair_kernel<40>": 1280.0 / 56.0, "blend_bwd_kernel<32>": 16.0}
# float atomics execute at the L2: ~20.4 G 64-byte requests/s chip-wide whatever the lanes of an instruction cover
# (tools/ubench_atomics.hip, profiles/r03_ubench_atomics.txt)
ATOMIC_REQUESTS_PEAK = 20.4e9


def build_roofline(args, kernels, n_vis, n_isect, ms_per_view, extra_view_bytes: int = 0, extra_note: str = ""):
    P = args.height * args.width
    pmc, pmc_file = load_pmc()
    per = {}
    for name, k in kernels.items():
        if not name.startswith("blend_") or "<" not in name:
            continue
        C = int(name[name.index("<") + 1:name.index(">")])
        fn = algorithmic_bytes_blend_bwd if "bwd" in name else algorithmic_bytes_blend_fwd
        b = fn(P, n_isect, n_vis, C)
        gbs = b / (k["avg_ms"] * 1e-3) / 1e9
        entry = {"achieved": gbs, "frac": gbs / HBM_PEAK_GBS, "algorithmic_bytes_per_launch": b,
                 "avg_launch_ms": k["avg_ms"], "launches": k["launches"], "total_ms": k["total_ms"]}
        rec = pmc.get(name)
        if isinstance(rec, dict):
            entry["traffic"] = rec.get("hbm_bytes")
            if rec.get("insts_valu"):
                w = float(rec.get("valu_issue_weight", 1.0))
                weighted = rec["insts_valu"] * w * 2.0            # cycles, 2 per plain wave64 instruction
                peak = SIMDS * CLOCK_HZ * k["avg_ms"] * 1e-3
                entry["valu"] = {"insts": rec["insts_valu"], "issue_weight": w,
                                 "weighted_issue_cycles": weighted, "peak_cycles": peak,
                                 "frac": weighted / peak}
                # fp32 MFMAs take the SAME issue budget as fp32 VALU instructions on gfx950 (tools/ubench_coexec.hip:
                # an MFMA wave beside a VALU wave on one SIMD takes the SUM of their times, not the maximum), so the
                # roof these kernels answer to is VALU cycles + matrix-pipe busy cycles per SIMD
                mf = float(rec.get("mfma_busy_cycles") or 0.0)
                if mf > 0.0:
                    n_mfma = mf / MFMA_CYCLES_F32.get(name, 32.0)      # the MFMA instructions are in SQ_INSTS_VALU too
                    valu_only = max(weighted - 2.0 * n_mfma, 0.0)
                    entry["fp32_issue"] = {"valu_cycles": valu_only, "mfma_busy_cycles": mf, "peak_cycles": peak,
                                           "frac": (valu_only + mf) / peak}
                if rec.get("valu_busy_pct") is not None:
                    entry["valu_busy_pct"] = rec["valu_busy_pct"]      # (the profiled launch's VALUBusy)
            if rec.get("atomic_requests_64B"):
                rate = rec["atomic_requests_64B"] / (k["avg_ms"] * 1e-3)
                entry["atomics"] = {"requests_64B": rec["atomic_requests_64B"], "achieved": rate,
                                    "peak": ATOMIC_REQUESTS_PEAK, "unit": "requests/s", "frac": rate / ATOMIC_REQUESTS_PEAK}
        elif rec is not None:
            entry["traffic"] = rec
        per[name] = entry
    if not per:
        return None
    dom = max(per, key=lambda n: per[n]["total_ms"])
    d = per[dom]
    N, K, D = args.points, 25, args.feature_dim
    b_view = algorithmic_bytes_whole_view(N, n_vis, n_isect, P, K, D, render_only=args.config == 5) + extra_view_bytes
    gbs_view = b_view / (ms_per_view * 1e-3) / 1e9
    # which roof does the dominant kernel answer to?  (the counters decide: VALUBusy / the issue fraction against the
    # HBM fraction; "hbm" stays the bound BASELINE.json prices the path against)
    fi = (d.get("fp32_issue") or {}).get("frac") or (d.get("valu") or {}).get("frac") or 0.0
    measured = "fp32_issue" if (fi > d["frac"] or (d.get("valu_busy_pct") or 0.0) > 60.0) else "hbm"
    return {"bound": "hbm", "bound_measured": measured, "kernel": dom, "achieved": d["achieved"], "peak": HBM_PEAK_GBS,
            "unit": "GB/s", "frac": d["frac"], "traffic": d.get("traffic"),
            "algorithmic_bytes_per_launch": d["algorithmic_bytes_per_launch"],
            "avg_launch_ms": d["avg_launch_ms"], "launches": d["launches"],
            "dominant_by": "total kernel time over the profiled step",
            "valu": d.get("valu"), "fp32_issue": d.get("fp32_issue"), "atomics": d.get("atomics"),
            "valu_busy_pct": d.get("valu_busy_pct"),
            "whole_view": {"bytes": b_view, "ms_per_view": ms_per_view, "achieved": gbs_view,
                           "frac": gbs_view / HBM_PEAK_GBS,
                           "note": "SURVEY 8d B_alg from the measured N_vis and I over wall time per view" + extra_note},
            "kernels": {k: {kk: vv for kk, vv in v.items() if kk != "launches"} for k, v in per.items()},
            "pmc_source": pmc_file, "pmc_commit": pmc.get("_commit") if isinstance(pmc, dict) else None,
            "note": "issue-bound kernels (DESIGN.md 3.5c/3.5d): `frac` is against HBM as BASELINE asks, "
                    "`valu.frac` against the VALU issue peak (1024 SIMDs x 2.4 GHz, 2 cycles per plain "
                    "wave64 instruction, DPP / permlane / transcendental weighted by measured cost), "
                    "`fp32_issue.frac` adds the matrix pipe's busy cycles (MFMAs and VALU instructions share one "
                    "issue budget per SIMD, tools/ubench_coexec.hip), `valu_busy_pct` is the profiler's VALUBusy, "
                    "`atomics.frac` the float-atomic request rate against the chip's ~20.4 G requests/s "
                    "(tools/ubench_atomics.hip).  traffic / valu / fp32_issue / atomics come from the committed "
                    "counter passes named in pmc_source, only the durations are measured live"}


# ------------------------------------------------------------------------------------------------
def run_rank(args, rank: int, local_rank: int, world: int) -> int:
    import torch
    import torch.distributed as dist

    selftest = args.ops != PRODUCT_OPS
    if args.device == "cuda":
        if not torch.cuda.is_available():
            raise SystemExit("bench.py needs an MI355X (no CPU fallback in the product path)")
        if args.share_gpu:
            if args.backend != "gloo":
                raise SystemExit("--share-gpu needs --backend gloo (RCCL refuses two ranks on one device)")
            local_rank = 0
        torch.cuda.set_device(local_rank)
        dev = torch.device("cuda", local_rank)
    else:
        if not selftest:
            raise SystemExit("--device cpu is a test hook and needs --ops <non-product module>: the "
                             "product operators have no CPU path")
        dev = torch.device("cpu")
    rccl_log = None
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl" and "NCCL_DEBUG" not in os.environ:
            # let RCCL say which algorithm / protocol it picks for the gradient messages (into a file of this rank's own;
            # rank 0's is parsed into the line's `comm.rccl` — whether the 472 MB ride a ring or the direct links is
            # what DESIGN 6's 6.0x-or-7.6x estimate turns on)
            rccl_log = "/tmp/gg_rccl_%d_rank%d.log" % (os.getpid(), rank)
            os.environ["NCCL_DEBUG"] = "INFO"
            os.environ.setdefault("NCCL_DEBUG_SUBSYS", "INIT,TUNING,GRAPH")
            os.environ["NCCL_DEBUG_FILE"] = rccl_log
        kw = {"device_id": dev} if (args.device == "cuda" and args.backend == "nccl") else {}
        dist.init_process_group(args.backend, rank=rank, world_size=world, **kw)

    if selftest:
        sys.path.insert(0, os.path.join(ROOT, "tests"))
    ops = importlib.import_module(args.ops)
    from gaussiangrasper_amd.camera import ring_cameras
    from gaussiangrasper_amd.dist import GradBucket, shard_views, train_step, train_step_pipelined
    from gaussiangrasper_amd.pipeline import backward_view, render_view, seeded_cotangents
    from gaussiangrasper_amd.scene import make_scene

    lib = None
    if not selftest:
        from gaussiangrasper_amd import _lib
        lib = _lib.load()

    def sync():
        if args.device == "cuda":
            torch.cuda.synchronize()

    def barrier():
        if world > 1:
            dist.barrier()
        sync()

    def max_over_ranks(x: float) -> float:
        if world == 1:
            return x
        t = torch.tensor([x], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    if args.config == 5:
        return run_render_only(args, rank, world, dev, ops, lib, barrier, max_over_ranks)

    scene_cpu = make_scene(args.points, feature_dim=args.feature_dim, config_index=3)
    if selftest:
        scene_cpu.scales.add_(1.6)      # tiny test images: splats large enough to cover pixels
    scene = scene_cpu.to(dev)
    total_views = args.views_per_step * world
    views = ring_cameras(total_views, args.height, args.width, device=dev)
    my_views = shard_views(total_views, rank, world)
    # The plugin route runs the CLASS train.sh loads: plugin.make_fused_model_class(...).get_outputs(camera) — camera
    # preparation from a `Cameras` object on the device (one host read-back of its scalars), operator sequence,
    # output dictionary — on gaussiangrasper_amd.stub's stand-ins for the two nerfstudio classes it touches
    # (nerfstudio is not installable here).  Its six Parameters are the scene's leaves: one bucket for both routes.
    from gaussiangrasper_amd.plugin import make_fused_model_class
    from gaussiangrasper_amd.stub import StubCameras, StubGaussianSplattingModel
    plugin_model = make_fused_model_class(StubGaussianSplattingModel, ops=ops)(scene)
    for k in ("means", "scales", "quats", "opacities", "colors_all", "feature"):
        setattr(scene, k, getattr(plugin_model, k))
    plugin_model.train()
    # cameras as FullImageDatamanager.next_train hands them out: on the device, stamped with their dataset index
    cameras = {v: StubCameras.from_view(views[v], device=dev, cam_idx=v) for v in my_views}
    bucket = GradBucket(scene.params())
    if args.deterministic:
        ops.set_deterministic_backward(True)
    direct = not args.no_direct and hasattr(ops, "register_grad_sink")
    if direct:
        # every parameter that enters an operator as a leaf: SH and feature rows always; means through the projection;
        # scales / quats / opacities through the plugin route's ActivateGaussians (the shim route's torch
        # activations leave those three to autograd)
        bucket.enable_direct(ops, defer_sh=True)   # + the SH gradients of a step's views expanded once (ShadeTail)
    # cotangents resident in HBM, one set reused for every view (dense N(0,1), seeded)
    probe = render_view(scene, views[my_views[0]], ops)
    cot = seeded_cotangents(probe, seed=1234)
    n_vis = int((probe["radii"] > 0).sum())
    n_isect = int(probe["num_tiles_hit"].long().sum())
    del probe
    if hasattr(ops, "clear_bin_cache"):
        ops.clear_bin_cache()

    def make_step(fused: bool):
        if fused:
            render = lambda v: plugin_model(cameras[v])
        else:
            render = lambda v: render_view(scene, views[v], ops, fused=False)

        def render_and_backward(v):
            backward_view(render(v), cot)
        render_and_backward.render = render
        render_and_backward.backward = lambda out: backward_view(out, cot)
        return render_and_backward

    overlap = not args.no_overlap
    stepper = None
    if args.reduce == "rs_ag":
        from gaussiangrasper_amd.dist import ShardedAdamStep, torch_adam_piece
        # the reference's per-group Adam hyper-parameters (method_configs.py:618-660), in scene.params() order
        lrs = (1.6e-4, 0.005, 0.001, 0.05, 5e-4, 5e-4)     # means, scales, quats, opacities, colors_all, feature
        stepper = ShardedAdamStep(bucket, [dict(lr=lr, eps=1e-15) for lr in lrs],
                                  adam_piece=torch_adam_piece if selftest else None)
    pipe_streams_box = [None]
    if torch.device(dev).type == "cuda" and not args.no_view_pipeline and not args.deterministic:
        pipe_streams_box[0] = [torch.cuda.Stream(device=dev, priority=int(p)) for p in args.pipe_priority.split(",")]

    def one_step(fn):
        red = stepper is None
        if pipe_streams_box[0] is not None:
            train_step_pipelined(fn.render, fn.backward, bucket, my_views, pipe_streams_box[0], reduce=red,
                                 overlap=overlap)
        else:
            train_step(fn, bucket, my_views, reduce=red, overlap=overlap)
        if stepper is not None:
            stepper.step()

    def timed(fn, steps):
        barrier()
        t0 = time.perf_counter()
        for _ in range(steps):
            one_step(fn)
        barrier()
        return max_over_ranks(time.perf_counter() - t0)

    def measure(fused: bool, warmup: int):
        """warm-up, the timed steps with the kernel timers OFF, then one profiled step"""
        fn = make_step(fused)
        for _ in range(warmup):
            one_step(fn)
        freeze_gc()
        t = timed(fn, args.steps)
        grads = bucket.gathered().detach().cpu().clone() if (args.dump_grads and rank == 0) else None
        kern, t_prof = {}, None
        if lib is not None and not args.no_prof:               # all ranks take part (collectives)
            saved, pipe_streams_box[0] = pipe_streams_box[0], None   # per-kernel times: one stream, no overlap
            one_step(fn)       # (untimed: the caching allocator's pool of THIS stream is empty after the pipelined steps,
            sync()             #  a first step on it pays for fresh hipMallocs — once 0.75 instead of 0.94 of wall)
            lib.gg_prof_reset()
            lib.gg_prof_enable(1)
            t_prof = timed(fn, PROFILED_STEPS)
            pipe_streams_box[0] = saved
            lib.gg_prof_enable(0)
            kern = read_kernel_times(lib)
        return t, kern, t_prof, grads

    # Two registration routes, both leave train.sh / render.sh untouched:
    #   plugin  NERFSTUDIO_METHOD_CONFIGS="gaussian-splatting=gaussiangrasper_amd.plugin:gaussian_splatting":
    #           the model subclass renders the four images of a view from ONE operator
    #           (ops.RasterizeSegments: one binning, one forward and one backward walk per colour array,
    #           one set of geometry gradients) — the headline;
    #   shim    PYTHONPATH=<repo>/shim only: the reference's model file with its four separate rasterize
    #           calls per view (they share the binning) — reported next to it.
    plugin_first = args.route == "plugin"
    bucket.comm_stats()                                       # (drop what the probe / set-up left)
    elapsed, kernels, t_prof, grads = measure(fused=plugin_first, warmup=args.warmup)
    comm = bucket.comm_stats() if world > 1 else None         # warm-up + timed + profiled steps of the headline route
    if args.torch_profile and rank == 0:
        from torch.profiler import ProfilerActivity, profile
        fn = make_step(plugin_first)
        with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as tp:
            one_step(fn)
            sync()
        with open(args.torch_profile, "w") as f:
            f.write(tp.key_averages(group_by_input_shape=True).table(sort_by="cuda_time_total", row_limit=60,
                                                                      max_name_column_width=60))
    if grads is not None:
        torch.save(grads, args.dump_grads)
    other = None
    if not args.no_secondary and not selftest:
        o_t, o_k, _, _ = measure(fused=not plugin_first, warmup=1)
        other = {"route": "shim (four rasterize calls per view)" if plugin_first else "plugin (one fused operator per view)",
                 "value": total_views * args.steps / o_t, "unit": "views/s",
                 "kernels": {k: round(v["avg_ms"], 4) for k, v in sorted(o_k.items())},
                 "kernel_ms_per_view": sum(v["total_ms"] for v in o_k.values()) / (args.views_per_step * PROFILED_STEPS)
                 if o_k else None}

    views_done = total_views * args.steps
    ms_per_view_rank = 1e3 * elapsed / (args.steps * args.views_per_step)
    result = {
        "metric": "rendered views/sec (fwd+bwd) at 1M Gaussians, 1600x1200, 32-ch feature",
        "value": views_done / elapsed, "unit": "views/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "f32",
        "data": "synthetic" if not selftest else f"selftest (non-product ops module {args.ops})",
        "config": {"workload": "BASELINE config 4 per-GPU share: %d Gaussians, %dx%d, SH deg 4 rgb + "
                               "%d-ch feature + depth + normal images per view (%s), "
                               "fwd+bwd, %d views/step/GPU, gradient all-reduce each step"
                               % (args.points, args.width, args.height, args.feature_dim,
                                  "nerfstudio plugin route: the timed callable is FusedGaussianSplattingModel."
                                  "get_outputs(camera) of gaussiangrasper_amd.plugin (the class train.sh loads, on "
                                  "stub.py's stand-ins for nerfstudio's base model and Cameras) — one fused rasterize "
                                  "operator per view"
                                  if plugin_first else "shim route: the reference's 4 rasterize calls per view",
                                  args.views_per_step),
                   "route": args.route, "deterministic_backward": bool(args.deterministic),
                   "python_gc": "as it comes" if os.environ.get("GG_BENCH_KEEP_GC") == "1" else
                                "gc.freeze() after the warm-up (bench.freeze_gc)",
                   "view_pipeline": ("backward of view k beside forward of view k + 1 on two streams"
                                     if pipe_streams_box[0] is not None else "off"),
                   "num_gaussians": args.points, "image": [args.height, args.width],
                   "feature_dim": args.feature_dim, "views_per_step_per_gpu": args.views_per_step,
                   "n_visible": n_vis, "num_intersects": n_isect,
                   "parallelism": f"view-parallel x{world}, replicated Gaussians",
                   "grad_allreduce_bytes": bucket.nbytes,
                   "grad_accumulation": "SH and feature gradients added into the step's gradient bucket by "
                                        "the backward kernels; the other parameters by autograd"
                                        if direct else "autograd",
                   "grad_allreduce": ("reduce-scatter + fused Adam on the shard + all-gather of the parameters "
                                      "(the step includes the optimizer)") if stepper is not None else
                                     ("per parameter, overlapped with the last view's backward"
                                      if overlap else "one collective after the last view"),
                   "backend": args.backend if world > 1 else None},
    }
    if rank == 0 and world > 1:
        # what the step waits for BEHIND its last kernel, and what RCCL chose (the first 8-GPU record should explain
        # itself: DESIGN 6 predicts ~5 ms exposed on a ring, ~0.75 ms on the direct links)
        result["comm"] = dict(comm or {}, backend=args.backend,
                              scheme=("reduce-scatter + sharded Adam + all-gather" if stepper is not None else
                                      ("all-reduce per parameter, armed on the last view" if overlap else
                                       "one all-reduce after the last view")),
                              rccl=parse_rccl_log(rccl_log))
    if rank == 0:
        result["roofline"] = build_roofline(args, kernels, n_vis, n_isect, ms_per_view_rank) \
            if kernels else None
        result["kernels"] = {k: round(v["avg_ms"], 4) for k, v in sorted(kernels.items())}
        if kernels:
            tot = sum(v["total_ms"] for v in kernels.values())
            result["kernel_time_fraction_of_wall"] = tot / (1e3 * t_prof)
            result["kernel_ms_per_view"] = tot / (args.views_per_step * PROFILED_STEPS)
        result["other_route"] = other
        result["cpu_baseline"] = None
        if world == 1 and not args.no_cpu_baseline and not selftest:
            try:
                result["cpu_baseline"] = cpu_baseline(args, scene_cpu, ring_cameras(
                    total_views, args.height, args.width)[0])
            except Exception as exc:  # noqa: BLE001
                result["cpu_baseline"] = {"error": repr(exc)}
    # BASELINE config 5 (render.sh's body at 5 M Gaussians, 128-dim feature, 1080p) rides along on the default
    # one-GPU run as a short object, so that the driver's line carries it: 5 steps of 2 views through the plugin's
    # model class in eval mode, its own kernel times, roofline and bounded CPU sample.  `--config 5` is the
    # full-length form.
    if world == 1 and not selftest and args.device == "cuda" and not args.no_config5:
        del plugin_model, bucket, scene, cot, cameras
        ops.clear_bin_cache()
        ops.clear_grad_sinks()
        torch.cuda.empty_cache()
        a5 = argparse.Namespace(**vars(args))
        a5.config, a5.points, a5.height, a5.width, a5.feature_dim = 5, 5_000_000, 1080, 1920, 128
        a5.views_per_step, a5.steps, a5.warmup = 2, 5, 1
        try:
            r5 = render_only_result(a5, rank, world, dev, ops, lib, barrier, max_over_ranks)
            result["config5"] = {k: r5[k] for k in ("metric", "value", "unit", "ms_per_view", "steps", "kernels",
                                                    "kernel_time_fraction_of_wall", "roofline", "cpu_baseline") if k in r5}
            result["config5"]["workload"] = r5["config"]["workload"]
            result["config5"]["num_intersects"] = r5["config"]["num_intersects"]
        except Exception as exc:  # noqa: BLE001
            result["config5"] = {"error": repr(exc)}
    if rank == 0:
        print(json.dumps(result), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    return 0


def render_only_result(args, rank, world, dev, ops, lib, barrier, max_over_ranks, with_cpu_baseline=True):
    """BASELINE config 5: the body of render.sh's loop (base_pipeline.py:401-408) per view —
    model(camera) in eval mode = project, SH, four rasterize forwards, then fea_up on every pixel of
    the feature image — on this rank's views; no gradient, no collective.  Returns the result object
    (rank 0; None on the other ranks)."""
    import torch
    from gaussiangrasper_amd.camera import ring_cameras
    from gaussiangrasper_amd.dist import shard_views
    from gaussiangrasper_amd.mlp import MLP
    from gaussiangrasper_amd.pipeline import render_view
    from gaussiangrasper_amd.scene import make_scene

    scene = make_scene(args.points, feature_dim=args.feature_dim, config_index=4).to(dev)
    torch.manual_seed(5)
    total_views = args.views_per_step * world
    views = ring_cameras(total_views, args.height, args.width, device=dev)
    my_views = shard_views(total_views, rank, world)

    fused = args.route == "plugin" and hasattr(ops, "rasterize_segments")
    model = None
    if fused:
        # the class render.sh loads (plugin.FusedGaussianSplattingModel on stub.py's stand-ins) in EVAL mode, as
        # eval_setup leaves it (utils/eval_utils.py:67-112): outputs = model(camera); model.fea_up(outputs["feature"])
        from gaussiangrasper_amd.plugin import make_fused_model_class
        from gaussiangrasper_amd.stub import StubCameras, StubGaussianSplattingModel
        model = make_fused_model_class(StubGaussianSplattingModel, ops=ops, fused_training=True)(scene).to(dev).eval()
        fea_up = model.fea_up
        cams = {v: StubCameras.from_view(views[v], device=dev) for v in my_views}
    else:
        fea_up = MLP(args.feature_dim, 512, hidden_list=[128]).to(dev)

    def render(v):
        # plugin route: the model class's get_outputs (ActivateGaussians, ShadeTail, one RasterizeSegments forward; the
        # visualisation images of :785-795 included, as the reference computes them); shim route: the reference's four
        # rasterize forwards
        if model is not None:
            out = model(cams[v])
        else:
            out = render_view(scene, views[v], ops, fused=False)
        clip = fea_up(out["feature"])
        return out, clip

    with torch.no_grad():
        out, _ = render(my_views[0])
        if model is not None:
            n_vis = int((model.radii > 0).sum())
            n_isect = int(ops.last_num_intersects() or 0)
        else:
            n_vis = int((out["radii"] > 0).sum())
            n_isect = int(out["num_tiles_hit"].long().sum())
        del out
        for _ in range(args.warmup):
            for v in my_views:
                render(v)
        freeze_gc()
        barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            for v in my_views:
                render(v)
        barrier()
        elapsed = max_over_ranks(time.perf_counter() - t0)
        kernels = {}
        if lib is not None and not args.no_prof:
            lib.gg_prof_reset()
            lib.gg_prof_enable(1)
            barrier()
            tp0 = time.perf_counter()
            for v in my_views:
                render(v)
            barrier()
            t_prof = max_over_ranks(time.perf_counter() - tp0)
            lib.gg_prof_enable(0)
            kernels = read_kernel_times(lib)
    views_done = total_views * args.steps
    ms_per_view = 1e3 * elapsed / (args.steps * args.views_per_step)
    result = {
        "metric": "rendered views/sec (render-only) at 5M Gaussians, 1920x1080, 128-ch feature + fea_up MLP",
        "value": views_done / elapsed, "unit": "views/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": "BASELINE config 5 per-GPU share (render.sh body): %d Gaussians, %dx%d, "
                               "SH deg 4 rgb + %d-ch feature + depth + normal forwards (%s) and the "
                               "fea_up MLP %d->128->512 on every pixel, render-only, %d views/step/GPU"
                               % (args.points, args.width, args.height, args.feature_dim,
                                  "plugin route: FusedGaussianSplattingModel.get_outputs in eval mode + model.fea_up, "
                                  "one fused rasterize operator" if fused
                                  else "shim route: the reference's 4 rasterize calls",
                                  args.feature_dim, args.views_per_step),
                   "route": "plugin" if fused else "shim",
                   "python_gc": "as it comes" if os.environ.get("GG_BENCH_KEEP_GC") == "1" else
                                "gc.freeze() after the warm-up (bench.freeze_gc)",
                   "num_gaussians": args.points, "image": [args.height, args.width],
                   "feature_dim": args.feature_dim, "views_per_step_per_gpu": args.views_per_step,
                   "n_visible": n_vis, "num_intersects": n_isect,
                   "parallelism": f"view-parallel x{world}, replicated Gaussians, no collective"},
    }
    if rank != 0:
        return None
    result["ms_per_view"] = ms_per_view
    mlp_bytes = 4 * 512 * args.height * args.width
    result["roofline"] = build_roofline(
        args, kernels, n_vis, n_isect, ms_per_view, extra_view_bytes=mlp_bytes,
        extra_note=" (render-only) + the fea_up MLP's 4 x 512 x P output bytes") if kernels else None
    result["kernels"] = {k: round(v["avg_ms"], 4) for k, v in sorted(kernels.items())}
    if kernels:
        result["kernel_time_fraction_of_wall"] = sum(v["total_ms"] for v in kernels.values()) / (1e3 * t_prof)
    result["cpu_baseline"] = None
    if world == 1 and with_cpu_baseline and not args.no_cpu_baseline:
        try:
            result["cpu_baseline"] = cpu_baseline_render(args, scene, views[my_views[0]], fea_up)
        except Exception as exc:                                 # the GPU numbers stand on their own
            result["cpu_baseline"] = {"value": None, "kind": "port", "sample": f"failed: {exc!r}"}
    return result


def run_render_only(args, rank, world, dev, ops, lib, barrier, max_over_ranks) -> int:
    import torch.distributed as dist
    result = render_only_result(args, rank, world, dev, ops, lib, barrier, max_over_ranks)
    if rank == 0:
        print(json.dumps(result), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    return 0


def main(argv=None) -> int:
    args = parse(argv)
    if "WORLD_SIZE" in os.environ:                      # started by torch.distributed.run (or by us)
        world = int(os.environ["WORLD_SIZE"])
        if world != args.gpus:
            print(f"bench.py: WORLD_SIZE={world} but --gpus {args.gpus}", file=sys.stderr)
            return 2
        return run_rank(args, int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")),
                        world)
    if args.gpus == 1:
        return run_rank(args, 0, 0, 1)
    # plain `python bench.py --gpus N`: start the N ranks ourselves.  Nothing in this process touches
    # the GPU (importing torch and counting devices do not initialise HIP), and the children are
    # fresh interpreters — no process that has initialised HIP is ever re-exec'd.
    from gaussiangrasper_amd.dist import spawn_ranks
    if args.device == "cuda" and not args.share_gpu:
        import torch
        n_dev = torch.cuda.device_count()               # does not initialise HIP on this image
        if n_dev < args.gpus:
            print(f"bench.py: --gpus {args.gpus} but only {n_dev} GPU(s) visible", file=sys.stderr)
            return 2
    child = [os.path.abspath(__file__)] + (sys.argv[1:] if argv is None else list(argv))
    return spawn_ranks(child, args.gpus)


if __name__ == "__main__":
    sys.exit(main())
