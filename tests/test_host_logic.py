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


_WORKER = r"""
import os, sys
sys.path[:0] = [{root!r}, {root!r} + '/shim', {root!r} + '/tests']
import torch, torch.distributed as dist
import oracle_ops
from gaussiangrasper_amd.camera import ring_cameras
from gaussiangrasper_amd.dist import GradBucket, shard_views, train_step
from gaussiangrasper_amd.pipeline import backward_view, render_view, seeded_cotangents
from gaussiangrasper_amd.scene import make_scene
dist.init_process_group('gloo', init_method='tcp://127.0.0.1:' + os.environ['GG_PORT'],
                        rank=int(os.environ['RANK']), world_size=int(os.environ['WORLD_SIZE']))
rank, world = dist.get_rank(), dist.get_world_size()
sc = make_scene(300, feature_dim=4, config_index=6); sc.scales.add_(1.6)
for p in sc.params(): p.requires_grad_(True)
views = ring_cameras(4, 32, 48)
bucket = GradBucket(sc.params())
def rb(v):
    out = render_view(sc, views[v], oracle_ops)
    backward_view(out, seeded_cotangents(out, seed=v))
train_step(rb, bucket, shard_views(len(views), rank, world))
if rank == 0:
    torch.save(bucket.flat.clone(), os.environ['GG_OUT'])
dist.barrier(); dist.destroy_process_group()
"""


def test_gradient_all_reduce_equals_single_process(tmp_path):
    """world_size 2 (gloo, CPU): views sharded round-robin, local accumulation, ONE all-reduce —
    the reduced gradient equals the single-process sum over all views"""
    import oracle_ops
    from gaussiangrasper_amd.camera import ring_cameras
    from gaussiangrasper_amd.dist import GradBucket, train_step
    from gaussiangrasper_amd.pipeline import backward_view, render_view, seeded_cotangents
    from gaussiangrasper_amd.scene import make_scene
    script = tmp_path / "worker.py"
    script.write_text(_WORKER.format(root=ROOT))
    out_file = tmp_path / "flat.pt"
    port = str(29500 + os.getpid() % 2000)
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", GG_PORT=port, GG_OUT=str(out_file),
                   OMP_NUM_THREADS="2")
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env))
    for p in procs:
        assert p.wait(timeout=300) == 0
    reduced = torch.load(out_file, weights_only=True)
    sc = make_scene(300, feature_dim=4, config_index=6)
    sc.scales.add_(1.6)
    for p in sc.params():
        p.requires_grad_(True)
    views = ring_cameras(4, 32, 48)
    bucket = GradBucket(sc.params())

    def rb(v):
        out = render_view(sc, views[v], oracle_ops)
        backward_view(out, seeded_cotangents(out, seed=v))
    train_step(rb, bucket, range(4), reduce=False)
    assert float(bucket.flat.abs().sum()) > 0
    assert torch.allclose(reduced, bucket.flat, rtol=1e-5, atol=1e-6 * float(bucket.flat.abs().max()))


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
