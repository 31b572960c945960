"""No-GPU tests of the host side: the gsplat-compatible shim surface and its error behaviour, the
scene / camera generators of SURVEY §8d, the reference call sequence replayed on the CPU through
oracle-backed operators (BASELINE config 1: 50k Gaussians, 400x300, RGB, "plumbing, no GPU"), and
the view-sharding + gradient all-reduce harness under gloo with world_size 2."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shim_resolves_the_reference_imports():
    # the five imports of nerfstudio/models/gaussian_splatting.py:46-50 (+ scripts/update.py:74)
    from gsplat._torch_impl import quat_to_rotmat  # noqa: F401
    from gsplat.nd_rasterize import NDRasterizeGaussians
    from gsplat.project_gaussians import ProjectGaussians
    from gsplat.rasterize import RasterizeGaussians
    from gsplat.sh import SphericalHarmonics, num_sh_bases  # noqa: F401
    for cls in (NDRasterizeGaussians, ProjectGaussians, RasterizeGaussians, SphericalHarmonics):
        assert issubclass(cls, torch.autograd.Function) and hasattr(cls, "apply")
    import gsplat
    assert gsplat.__version__.startswith("0.1.0")


def test_product_has_no_cpu_fallback_and_never_imports_the_oracle():
    from gsplat.project_gaussians import ProjectGaussians
    from gsplat.rasterize import RasterizeGaussians
    n = 3
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ProjectGaussians.apply(torch.zeros(n, 3), torch.ones(n, 3), 1, torch.ones(n, 4), torch.eye(4)[:3],
                               torch.eye(4), 10.0, 10.0, 8.0, 8.0, 16, 16, (1, 1, 1))
    zi = torch.zeros(n, dtype=torch.int32)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        RasterizeGaussians.apply(torch.zeros(n, 2), torch.zeros(n), zi, torch.zeros(n, 3), zi,
                                 torch.zeros(n, 3), torch.zeros(n, 1), 16, 16, torch.zeros(3))
    # static check: nothing under the product package or the shim mentions the oracle package
    for base in ("gaussiangrasper_amd", "shim"):
        for dp, _, files in os.walk(os.path.join(ROOT, base)):
            for f in files:
                if f.endswith((".py", ".hip", ".h")):
                    src = open(os.path.join(dp, f)).read()
                    assert "import oracle" not in src and "from oracle" not in src, (dp, f)
                    assert "libgg_oracle" not in src, (dp, f)


def test_shape_errors_match_gsplat_conventions():
    """ValueError / assert on bad shapes, raised before any device work"""
    from gsplat.nd_rasterize import NDRasterizeGaussians
    from gsplat.project_gaussians import ProjectGaussians
    from gsplat.rasterize import RasterizeGaussians
    from gsplat.sh import SphericalHarmonics
    n = 4
    f = torch.zeros
    zi = torch.zeros(n, dtype=torch.int32)
    with pytest.raises(ValueError, match=r"colors must have dimensions \(N, 3\)"):
        RasterizeGaussians.apply(f(n, 2), f(n), zi, f(n, 3), zi, f(n, 4), f(n, 1), 16, 16, f(4))
    with pytest.raises(ValueError, match=r"opacity must have dimensions \(N, 1\)"):
        RasterizeGaussians.apply(f(n, 2), f(n), zi, f(n, 3), zi, f(n, 3), f(n), 16, 16, f(3))
    with pytest.raises(ValueError, match=r"xys must have dimensions \(N, 2\)"):
        NDRasterizeGaussians.apply(f(n, 3), f(n), zi, f(n, 3), zi, f(n, 8), f(n, 1), 16, 16, f(8))
    with pytest.raises(AssertionError, match="background"):
        NDRasterizeGaussians.apply(f(n, 2), f(n), zi, f(n, 3), zi, f(n, 8), f(n, 1), 16, 16, f(3))
    with pytest.raises(ValueError):
        ProjectGaussians.apply(f(n, 2), f(n, 3), 1, f(n, 4), torch.eye(4)[:3], torch.eye(4), 1.0, 1.0,
                               0.0, 0.0, 16, 16, (1, 1, 1))
    with pytest.raises(AssertionError):
        SphericalHarmonics.apply(3, f(n, 3), f(n, 9, 3))      # 9 bases cannot serve degree 3
    with pytest.raises(ValueError):
        SphericalHarmonics.apply(1, f(n, 3), f(n, 5, 3))      # 5 is not a valid basis count


def test_scene_and_cameras_are_deterministic():
    from gaussiangrasper_amd.camera import ring_cameras
    from gaussiangrasper_amd.scene import make_scene
    a, b = make_scene(1000, config_index=3), make_scene(1000, config_index=3)
    for x, y in zip(a.params(), b.params()):
        assert torch.equal(x, y)
    assert not torch.equal(a.means, make_scene(1000, config_index=4).means)
    assert a.colors_all.shape == (1000, 25, 3) and a.feature.shape == (1000, 32)
    assert a.means.abs().max() <= 1 and a.means[:, 2].abs().max() <= 0.5
    ratio = (a.scales.max(-1).values - a.scales.min(-1).values).exp()
    assert ratio.max() <= 10.0 + 1e-4                                   # max_gauss_ratio
    assert torch.allclose(a.quats.norm(dim=-1), torch.ones(1000), atol=1e-6)
    v = ring_cameras(8, 1200, 1600)
    assert len(v) == 8 and v[0].tile_bounds == (100, 75, 1)
    assert abs(v[0].fx - 0.5 * 1600 / np.tan(np.pi / 6)) < 1e-3 and v[0].cx == 800 and v[0].cy == 600
    assert abs(float(v[3].cam_pos.norm()) - 2.5) < 1e-5


def test_reference_call_sequence_on_cpu_config1():
    """BASELINE config 1: 50k random Gaussians, one 400x300 camera, RGB only, through the same
    render_view() the GPU path uses, with the oracle-backed operators standing in on the CPU."""
    import oracle_ops
    from gaussiangrasper_amd.camera import ring_cameras
    from gaussiangrasper_amd.pipeline import backward_view, render_view, seeded_cotangents
    from gaussiangrasper_amd.scene import make_scene
    sc = make_scene(50_000, config_index=0)
    for p in sc.params():
        p.requires_grad_(True)
    out = render_view(sc, ring_cameras(1, 300, 400)[0], oracle_ops, channels=("rgb",))
    assert out["rgb"].shape == (300, 400, 3) and 0.0 <= float(out["rgb"].detach().min()) and float(out["rgb"].detach().max()) <= 1.0
    assert int((out["radii"] > 0).sum()) > 45_000
    backward_view(out, seeded_cotangents(out))
    assert out["xys"].grad is not None and float(out["xys"].grad.abs().sum()) > 0
    for name, p in zip(("means", "scales", "quats", "opacities", "colors_all"), sc.params()):
        assert p.grad is not None and torch.isfinite(p.grad).all() and float(p.grad.abs().sum()) > 0, name
    assert sc.feature.grad is None                                      # feature not rendered here


def test_all_four_outputs_and_xys_grad_accumulates():
    import oracle_ops
    from gaussiangrasper_amd.camera import ring_cameras
    from gaussiangrasper_amd.pipeline import render_view
    from gaussiangrasper_amd.scene import make_scene
    sc = make_scene(800, feature_dim=6, config_index=2)
    sc.scales.add_(1.5)
    for p in sc.params():
        p.requires_grad_(True)
    v = ring_cameras(2, 48, 64)[0]
    out = render_view(sc, v, oracle_ops)
    assert out["rgb"].shape == (48, 64, 3) and out["feature"].shape == (48, 64, 6)
    assert out["depth"].shape == (48, 64, 1) and out["normal"].shape == (48, 64, 3)
    empty = out["depth"][..., 0] > 9.99       # background depth 10 where nothing was hit
    assert 0 < int(empty.sum()) < 48 * 64
    g_each = []
    for k in ("rgb", "feature", "depth", "normal"):
        (g,) = torch.autograd.grad(out[k].sum(), out["xys"], retain_graph=True)
        g_each.append(g)
    out["xys"].grad = None   # retain_grad hooks also fired during the autograd.grad calls above
    total = sum(out[k].sum() for k in ("rgb", "feature", "depth", "normal"))
    total.backward()
    assert torch.allclose(out["xys"].grad, sum(g_each), rtol=1e-5, atol=1e-6)


def test_fused_single_call_path_matches_four_calls_on_cpu():
    """SURVEY 8f-1 host logic: feature|rgb|depth|normal through ONE ND call (oracle-backed on CPU)
    gives the four images of the four separate calls and the same parameter gradients."""
    import oracle_ops
    from gaussiangrasper_amd.camera import ring_cameras
    from gaussiangrasper_amd.pipeline import backward_view, render_view, seeded_cotangents
    from gaussiangrasper_amd.scene import make_scene
    v = ring_cameras(2, 48, 64)[1]
    outs, grads = [], []
    for fused in (False, True):
        sc = make_scene(600, feature_dim=5, config_index=8)
        sc.scales.add_(1.5)
        for p in sc.params():
            p.requires_grad_(True)
        out = render_view(sc, v, oracle_ops, fused=fused)
        backward_view(out, seeded_cotangents(out, seed=5))
        outs.append(out)
        grads.append([p.grad.clone() for p in sc.params()])
    for k in ("rgb", "feature", "depth", "normal"):
        assert outs[0][k].shape == outs[1][k].shape
        assert torch.equal(outs[0][k], outs[1][k]), k       # same per-channel fmaf sequence
    for a, b in zip(grads[0], grads[1]):
        assert torch.allclose(a, b, rtol=1e-4, atol=1e-6 * float(a.abs().max()))


def test_view_sharding():
    from gaussiangrasper_amd.dist import shard_views
    assert shard_views(64, 3, 8) == list(range(3, 64, 8))
    allv = sorted(v for r in range(8) for v in shard_views(64, r, 8))
    assert allv == list(range(64))
    assert shard_views(3, 5, 8) == []


def _run_bench(extra, tmp_path, gpus, env_extra=None, timeout=600):
    """Launch bench.py the way the driver does (`python bench.py --gpus N ...`, no WORLD_SIZE in the
    environment): for N > 1 it must create its own N ranks."""
    env = {k: v for k, v in os.environ.items()
           if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env["OMP_NUM_THREADS"] = "2"
    if env_extra:
        env.update(env_extra)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(gpus), "--backend", "gloo",
           "--device", "cpu", "--ops", "oracle_ops", "--points", "300", "--height", "32", "--width", "48",
           "--feature-dim", "4", "--views-per-step", "2", "--steps", "1", "--warmup", "1",
           "--no-cpu-baseline", "--no-secondary"] + extra
    return subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=timeout)


def _json_lines(stdout):
    import json
    return [json.loads(ln) for ln in stdout.splitlines() if ln.startswith("{")]


def test_bench_spawns_its_own_ranks_and_reduces_gradients(tmp_path):
    """`python bench.py --gpus 2` (gloo, CPU test hook, oracle-backed operators) starts two ranks by
    itself, prints ONE line with n_gpus 2, and the reduced gradient equals the single-process sum
    over the same four views — with the per-parameter overlapped reduction and with the single
    collective."""
    g2, g2b, g1 = tmp_path / "g2.pt", tmp_path / "g2b.pt", tmp_path / "g1.pt"
    r2 = _run_bench(["--dump-grads", str(g2)], tmp_path, 2)
    assert r2.returncode == 0, r2.stderr[-2000:]
    lines = _json_lines(r2.stdout)
    assert len(lines) == 1 and lines[0]["n_gpus"] == 2
    assert lines[0]["data"].startswith("selftest")
    assert lines[0]["config"]["backend"] == "gloo"
    nbytes = lines[0]["config"]["grad_allreduce_bytes"]
    assert nbytes == 4 * 300 * (3 + 3 + 4 + 1 + 75 + 4)
    # the N > 1 line explains its communication itself: what the step waited for behind its last
    # kernel, how many bytes in how many messages, by which scheme; `rccl` is RCCL's own account (none under gloo)
    comm = lines[0]["comm"]
    assert comm["bytes_per_step"] == nbytes and comm["collectives_per_step"] == 6 and comm["backend"] == "gloo"
    assert comm["steps"] >= lines[0]["steps"] and comm["exposed_ms_mean"] >= 0.0
    assert comm["exposed_ms_max"] >= comm["exposed_ms_mean"] and comm["rccl"] is None
    assert comm["scheme"].startswith("all-reduce per parameter")
    r2b = _run_bench(["--dump-grads", str(g2b), "--no-overlap"], tmp_path, 2)
    assert r2b.returncode == 0, r2b.stderr[-2000:]
    # single process, same 4 views: 2 ranks x 2 views/step == 1 rank x 4 views/step
    r1 = _run_bench(["--dump-grads", str(g1), "--views-per-step", "4"], tmp_path, 1)
    assert r1.returncode == 0, r1.stderr[-2000:]
    l1 = _json_lines(r1.stdout)
    assert len(l1) == 1 and l1[0]["n_gpus"] == 1 and l1[0]["config"]["grad_allreduce_bytes"] == nbytes
    assert "comm" not in l1[0]
    a, b, c = (torch.load(f, weights_only=True) for f in (g2, g2b, g1))
    assert float(c.abs().sum()) > 0
    tol = 1e-6 * float(c.abs().max())
    assert torch.allclose(a, c, rtol=1e-5, atol=tol)
    assert torch.allclose(b, c, rtol=1e-5, atol=tol)


def test_bench_reduce_scatter_adam_all_gather_scheme_runs_on_two_gloo_ranks(tmp_path):
    """`--reduce rs_ag`: the step ends with reduce-scatter, Adam on the rank's shard and an all-gather of the
    parameters (dist.ShardedAdamStep; its arithmetic is pinned in tests/test_sharded_step.py)"""
    r = _run_bench(["--reduce", "rs_ag"], tmp_path, 2)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = _json_lines(r.stdout)
    assert len(lines) == 1 and lines[0]["n_gpus"] == 2
    assert lines[0]["config"]["grad_allreduce"].startswith("reduce-scatter")


def test_bench_refuses_a_world_size_that_is_not_gpus(tmp_path):
    """no silent single-GPU fallback: WORLD_SIZE=1 with --gpus 2 is an error, not a 1-GPU run"""
    r = _run_bench([], tmp_path, 2, env_extra={"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode != 0 and "WORLD_SIZE=1" in r.stderr
    assert not _json_lines(r.stdout)
    # and the product operators cannot be benchmarked on the CPU
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--device", "cpu", "--points", "10"],
                       env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "no CPU path" in r.stderr


def test_grad_bucket_rebind_after_parameter_replacement():
    """densification replaces the parameter tensors (ref gaussian_splatting.py:434-439): rebind()
    re-aliases .grad, and an armed hook on a stale alias raises instead of reducing garbage"""
    from gaussiangrasper_amd.dist import GradBucket
    ps = [torch.zeros(5, 3, requires_grad=True), torch.zeros(5, 1, requires_grad=True)]
    b = GradBucket(ps)
    (ps[0].sum() * 2 + ps[1].sum() * 3).backward()
    assert torch.equal(b.gathered(), torch.cat([torch.full((15,), 2.0), torch.full((5,), 3.0)]))
    assert all((sl.data_ptr() - b.flat.data_ptr()) % 256 == 0 for sl in b.slices)
    ps2 = [torch.zeros(7, 3, requires_grad=True), torch.zeros(7, 1, requires_grad=True)]
    b.rebind(ps2)
    assert b.payload == 28 and b.nbytes == 112
    (ps2[0].sum() + ps2[1].sum() * 5).backward()
    assert torch.equal(b.gathered(), torch.cat([torch.ones(21), torch.full((7,), 5.0)]))
    ps2[0].grad = torch.zeros(7, 3)        # somebody broke the alias
    b.arm()
    with pytest.raises(RuntimeError, match="no longer aliases"):
        ps2[0].sum().backward()


def test_checkpoint_and_ply_round_trips(tmp_path):
    """SURVEY 8f-4 wire formats: the reference's checkpoint layout (trainer.py:437-449, model keys
    gaussian_splatting.py:271-281 + fea_up) and its splat PLY property set (exporter.py:499-525)."""
    from gaussiangrasper_amd import interop
    from gaussiangrasper_amd.scene import make_scene
    sc = make_scene(257, feature_dim=32, sh_degree=4, config_index=7)
    mlp_state = {"layers.0.weight": torch.randn(128, 32), "layers.0.bias": torch.randn(128),
                 "layers.2.weight": torch.randn(512, 128), "layers.2.bias": torch.randn(512)}
    ck = tmp_path / "step-000001234.ckpt"
    interop.save_checkpoint(ck, sc, mlp_state, step=1234)
    blob = torch.load(ck, weights_only=True)          # what the reference's trainer reads
    assert blob["step"] == 1234 and set(blob) == {"step", "pipeline", "optimizers", "schedulers", "scalers"}
    assert sorted(blob["pipeline"]) == sorted(
        ["_model." + k for k in interop.PARAM_KEYS + interop.MLP_KEYS])
    sc2, mlp2, step = interop.load_checkpoint(ck)
    assert step == 1234
    for a, b in zip(sc.params(), sc2.params()):
        assert torch.equal(a, b)
    assert all(torch.equal(mlp_state[k], mlp2[k]) for k in mlp_state)
    with pytest.raises(KeyError):
        interop.scene_from_state_dict({"_model.means": sc.means})
    bad = interop.state_dict_from_scene(sc)
    bad["_model.quats"] = torch.zeros(257, 3)
    with pytest.raises(ValueError):
        interop.scene_from_state_dict(bad)

    ply = tmp_path / "point_cloud.ply"
    interop.export_ply(ply, sc)
    head = open(ply, "rb").read(4096).split(b"end_header\n")[0].decode().splitlines()
    assert head[:3] == ["ply", "format binary_little_endian 1.0", "element vertex 257"]
    names = [ln.split()[2] for ln in head[3:]]
    assert names == (["x", "y", "z", "nx", "ny", "nz", "red", "green", "blue", "f_dc_0", "f_dc_1", "f_dc_2"]
                     + [f"f_rest_{i}" for i in range(72)] + ["opacity", "scale_0", "scale_1", "scale_2",
                                                             "rot_0", "rot_1", "rot_2", "rot_3"])
    assert os.path.getsize(ply) == len("\n".join(head)) + len("\nend_header\n") + 257 * (4 * (6 + 3 + 72 + 1 + 3 + 4) + 3)
    sc3 = interop.load_ply(ply, feature_dim=32)
    for name in ("means", "scales", "quats", "opacities"):
        assert torch.equal(getattr(sc, name), getattr(sc3, name)), name
    assert torch.equal(sc.colors_all[:, 1:], sc3.colors_all[:, 1:])
    # f_dc holds SH2RGB(dc) as in the reference's exporter: back through RGB2SH within rounding
    assert torch.allclose(sc.colors_all[:, 0], sc3.colors_all[:, 0], atol=1e-6)
    assert sc3.feature.shape == (257, 32) and not sc3.feature.any()


def test_load_ply_accepts_the_reference_exporters_literal_property_set(tmp_path):
    """The reference's exporter reshapes shs_rest to (N, 72, 1) and loops over shape[-1] == 1
    (scripts/exporter.py:508-512): its PLY carries exactly one higher-band property, f_rest_0.
    A hand-built file with that property set (in a different order, as open3d may write it) loads,
    with the missing bands zero-filled."""
    from gaussiangrasper_amd import interop
    from gaussiangrasper_amd.constants import SH_C0
    n = 5
    rng = np.random.default_rng(3)
    props = [("x", "<f4"), ("y", "<f4"), ("z", "<f4"), ("nx", "<f4"), ("ny", "<f4"), ("nz", "<f4"),
             ("red", "u1"), ("green", "u1"), ("blue", "u1"), ("opacity", "<f4"), ("f_rest_0", "<f4"),
             ("rot_0", "<f4"), ("rot_1", "<f4"), ("rot_2", "<f4"), ("rot_3", "<f4"),
             ("scale_0", "<f4"), ("scale_1", "<f4"), ("scale_2", "<f4"),
             ("f_dc_0", "<f4"), ("f_dc_1", "<f4"), ("f_dc_2", "<f4")]
    rec = np.zeros(n, dtype=np.dtype(props))
    for name, dt in props:
        rec[name] = rng.random(n).astype(np.float32) if dt == "<f4" else rng.integers(0, 255, n)
    header = ["ply", "format binary_little_endian 1.0", "comment hand-built", f"element vertex {n}"]
    header += [f"property {'float' if d == '<f4' else 'uchar'} {name}" for name, d in props] + ["end_header"]
    ply = tmp_path / "ref.ply"
    ply.write_bytes(("\n".join(header) + "\n").encode("ascii") + rec.tobytes())
    sc = interop.load_ply(ply, feature_dim=32)
    assert sc.colors_all.shape == (n, 25, 3)
    assert np.array_equal(sc.colors_all[:, 1, 0].numpy(), rec["f_rest_0"])
    rest = sc.colors_all[:, 1:].reshape(n, -1)
    assert not rest[:, 1:].any()
    assert np.allclose(sc.colors_all[:, 0, 1].numpy(), (rec["f_dc_1"] - 0.5) / SH_C0, atol=1e-6)
    assert np.array_equal(sc.means[:, 2].numpy(), rec["z"])
    assert np.array_equal(sc.quats[:, 3].numpy(), rec["rot_3"])
    # our writer's `reference_literal` mode produces that property set
    out = tmp_path / "lit.ply"
    interop.export_ply(out, sc, reference_literal=True)
    head = open(out, "rb").read(4096).split(b"end_header\n")[0].decode().splitlines()
    names = [ln.split()[2] for ln in head[3:]]
    assert [x for x in names if x.startswith("f_rest_")] == ["f_rest_0"]
    sc2 = interop.load_ply(out)
    assert torch.equal(sc2.colors_all[:, 1, 0], sc.colors_all[:, 1, 0]) and torch.equal(sc2.means, sc.means)
