"""One view through the operator sequence of the reference model's `get_outputs`
(nerfstudio/models/gaussian_splatting.py:699-784): project -> SH -> rasterize rgb (3 ch),
feature (D ch, ND op), depth (3 replicated ch, background 10, channel 0 kept), normal (3 ch).
Used by bench.py, the multi-GPU harness and the tests; `ops` is any namespace that exposes the
four gsplat-style autograd.Functions + quat_to_rotmat (the product: gaussiangrasper_amd.ops)."""
from __future__ import annotations

from typing import Dict

import torch

from .camera import ViewParams
from .scene import Scene


def smallest_axis_normals(quats: torch.Tensor, log_scales: torch.Tensor, quat_to_rotmat) -> torch.Tensor:
    """Column of R(q) along the smallest scale (reference get_smallest_axis, :605-619)."""
    rot = quat_to_rotmat(quats)
    idx = log_scales.exp().min(dim=-1)[1][..., None, None].expand(-1, 3, -1)
    return rot.gather(2, idx).squeeze(dim=2)


def render_view(scene: Scene, view: ViewParams, ops, sh_degree_to_use: int = 4,
                channels=("rgb", "feature", "depth", "normal")) -> Dict[str, torch.Tensor]:
    dev = scene.means.device
    h, w = view.height, view.width
    xys, depths, radii, conics, num_tiles_hit, cov3d = ops.ProjectGaussians.apply(
        scene.means, torch.exp(scene.scales), 1,
        scene.quats / scene.quats.norm(dim=-1, keepdim=True),
        view.viewmat[:3, :], view.projmat, view.fx, view.fy, view.cx, view.cy, h, w,
        view.tile_bounds)
    out: Dict[str, torch.Tensor] = {"xys": xys, "radii": radii, "depths": depths, "conics": conics,
                                    "num_tiles_hit": num_tiles_hit}
    if xys.requires_grad:
        xys.retain_grad()           # densification statistics read xys.grad (:724-725, :376-393)
    if "rgb" in channels:
        viewdirs = scene.means.detach() - view.cam_pos
        viewdirs = viewdirs / viewdirs.norm(dim=-1, keepdim=True)
        rgbs = ops.SphericalHarmonics.apply(sh_degree_to_use, viewdirs, scene.colors_all)
        rgbs = torch.clamp(rgbs + 0.5, 0.0, 1.0)
        out["rgb"] = ops.RasterizeGaussians.apply(
            xys, depths, radii, conics, num_tiles_hit, rgbs, torch.sigmoid(scene.opacities), h, w,
            torch.zeros(3, device=dev))
    if "feature" in channels:
        out["feature"] = ops.NDRasterizeGaussians.apply(
            xys, depths, radii, conics, num_tiles_hit, scene.feature,
            torch.sigmoid(scene.opacities), h, w, torch.zeros(scene.feature.shape[1], device=dev))
    if "depth" in channels:
        out["depth"] = ops.RasterizeGaussians.apply(
            xys, depths, radii, conics, num_tiles_hit, depths[:, None].repeat(1, 3),
            torch.sigmoid(scene.opacities), h, w, torch.ones(3, device=dev) * 10)[..., 0:1]
    if "normal" in channels:
        normals = smallest_axis_normals(scene.quats, scene.scales, ops.quat_to_rotmat)
        out["normal"] = ops.RasterizeGaussians.apply(
            xys, depths, radii, conics, num_tiles_hit, normals, torch.sigmoid(scene.opacities),
            h, w, torch.zeros(3, device=dev))
    return out


def seeded_cotangents(outputs: Dict[str, torch.Tensor], seed: int = 0) -> Dict[str, torch.Tensor]:
    """Dense N(0,1) v_out for every image output, generated on the CPU so that CPU
    and GPU runs see the same numbers."""
    g = torch.Generator(device="cpu").manual_seed(seed)
    cot = {}
    for name in ("rgb", "feature", "depth", "normal"):
        if name in outputs:
            t = outputs[name]
            cot[name] = torch.randn(t.shape, generator=g, dtype=torch.float32).to(t.device)
    return cot


def backward_view(outputs: Dict[str, torch.Tensor], cotangents: Dict[str, torch.Tensor]) -> None:
    """loss = sum_k <output_k, v_out_k>; one backward through all rasterize calls, SH and project."""
    loss = None
    for name, v in cotangents.items():
        term = (outputs[name] * v).sum()
        loss = term if loss is None else loss + term
    loss.backward()
