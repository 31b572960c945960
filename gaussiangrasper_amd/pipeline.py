"""One view through the operator sequence of the reference model's `get_outputs`
(nerfstudio/models/gaussian_splatting.py:699-784): project -> SH -> rasterize rgb (3 ch),
feature (D ch, ND op), depth (3 replicated ch, background 10, channel 0 kept), normal (3 ch).
Used by bench.py, the multi-GPU harness and the tests; `ops` is any namespace that exposes the
four gsplat-style autograd.Functions + quat_to_rotmat (the product: gaussiangrasper_amd.ops).

Two stages so that tests can feed bit-identical activated inputs to two operator stacks:
  activate()            the caller-side torch elementwise work: exp(scales),
                        q/|q|, sigmoid(opacities) (recomputed per rasterize call, as the reference
                        does), view directions, smallest-axis normals;
  rasterize_activated() the operator calls themselves, argument for argument as the reference."""
from __future__ import annotations

from typing import Callable, Dict, Union

import torch

from .camera import ViewParams
from .scene import Scene

CHANNELS = ("rgb", "feature", "depth", "normal")


def smallest_axis_normals(quats: torch.Tensor, log_scales: torch.Tensor, quat_to_rotmat) -> torch.Tensor:
    """Column of R(q) along the smallest scale (reference get_smallest_axis, :605-619)."""
    rot = quat_to_rotmat(quats)
    idx = log_scales.exp().min(dim=-1)[1][..., None, None].expand(-1, 3, -1)
    return rot.gather(2, idx).squeeze(dim=2)


def activate(scene: Scene, view: ViewParams, quat_to_rotmat) -> Dict[str, Union[torch.Tensor, Callable]]:
    viewdirs = scene.means.detach() - view.cam_pos.to(scene.means.device)
    viewdirs = viewdirs / viewdirs.norm(dim=-1, keepdim=True)
    return {
        "means": scene.means,
        "scales": torch.exp(scene.scales),
        "quats": scene.quats / scene.quats.norm(dim=-1, keepdim=True),
        "opac": lambda: torch.sigmoid(scene.opacities),     # :742,:754,:766,:780 — one per call
        "viewdirs": viewdirs,
        "sh": scene.colors_all,
        "feature": scene.feature,
        "normals": smallest_axis_normals(scene.quats, scene.scales, quat_to_rotmat),
    }


def activate_fused(scene: Scene, view: ViewParams, ops) -> Dict[str, Union[torch.Tensor, Callable]]:
    """`activate` through ONE operator (ops.ActivateGaussians: one kernel forward, one backward) where the
    operator namespace has it; same dictionary, opacity activated once."""
    if not hasattr(ops, "ActivateGaussians"):
        return activate(scene, view, ops.quat_to_rotmat)
    scales, quats_n, opac, viewdirs, normals = ops.ActivateGaussians.apply(
        scene.means, scene.scales, scene.quats, scene.opacities, view.cam_pos.to(scene.means.device))
    return {"means": scene.means, "scales": scales, "quats": quats_n, "opac": opac, "viewdirs": viewdirs,
            "sh": scene.colors_all, "feature": scene.feature, "normals": normals}


def rasterize_activated(act: Dict, view: ViewParams, ops, sh_degree_to_use: int = 4,
                        channels=CHANNELS) -> Dict[str, torch.Tensor]:
    dev = act["means"].device
    h, w = view.height, view.width
    opac = (lambda: act["opac"]()) if callable(act["opac"]) else (lambda: act["opac"])
    xys, depths, radii, conics, num_tiles_hit, cov3d = ops.ProjectGaussians.apply(
        act["means"], act["scales"], 1, act["quats"], view.viewmat[:3, :].to(dev),
        view.projmat.to(dev), view.fx, view.fy, view.cx, view.cy, h, w, view.tile_bounds)
    out: Dict[str, torch.Tensor] = {"xys": xys, "radii": radii, "depths": depths, "conics": conics,
                                    "num_tiles_hit": num_tiles_hit}
    if xys.requires_grad:
        xys.retain_grad()           # densification statistics read xys.grad (:724-725, :376-393)
    if "rgb" in channels:
        rgbs = ops.SphericalHarmonics.apply(sh_degree_to_use, act["viewdirs"], act["sh"])
        rgbs = torch.clamp(rgbs + 0.5, 0.0, 1.0)
        out["rgb"] = ops.RasterizeGaussians.apply(
            xys, depths, radii, conics, num_tiles_hit, rgbs, opac(), h, w, torch.zeros(3, device=dev))
    if "feature" in channels:
        out["feature"] = ops.NDRasterizeGaussians.apply(
            xys, depths, radii, conics, num_tiles_hit, act["feature"], opac(), h, w,
            torch.zeros(act["feature"].shape[1], device=dev))
    if "depth" in channels:
        out["depth"] = ops.RasterizeGaussians.apply(
            xys, depths, radii, conics, num_tiles_hit, depths[:, None].repeat(1, 3), opac(), h, w,
            torch.ones(3, device=dev) * 10)[..., 0:1]
    if "normal" in channels:
        out["normal"] = ops.RasterizeGaussians.apply(
            xys, depths, radii, conics, num_tiles_hit, act["normals"], opac(), h, w,
            torch.zeros(3, device=dev))
    return out


_CONSTANTS: Dict = {}


def _zeros(d: int, dev) -> torch.Tensor:
    """Cached constant backgrounds (read-only): no fill kernel per view."""
    key = ("zeros", d, str(dev))
    if key not in _CONSTANTS:
        _CONSTANTS[key] = torch.zeros(d, device=dev)
    return _CONSTANTS[key]


def _tail_background(dev) -> torch.Tensor:
    key = ("tail", str(dev))
    if key not in _CONSTANTS:
        bg = torch.zeros(7, device=dev)
        bg[3] = 10.0                      # depth background (reference :769)
        _CONSTANTS[key] = bg
    return _CONSTANTS[key]


def fused_images(ops, xys, depths, radii, conics, num_tiles_hit, opac, h, w, feature, rgbs, normals, tail=None,
                 packed=None):
    """feature (D) | rgb (3) | depth (1, background 10) | normal (3) images from one binning.
    With `ops.rasterize_segments` (the product) the feature array and the 7-channel rgb|depth|normal array
    are two segments of ONE operator: no (N, D+7) concatenation, aligned feature rows, one set of
    geometry gradients.  Otherwise (the oracle-backed test operators) one NDRasterize call on the
    concatenation.  Same images bit for bit either way."""
    dev = xys.device
    d = feature.shape[1]
    if hasattr(ops, "rasterize_segments"):
        if tail is None:
            tail = torch.cat([rgbs, depths[:, None], normals], dim=1)
        bg_tail = _tail_background(dev)
        kw = {"packed": packed} if packed is not None else {}
        feat_im, rgb, depth, normal = ops.rasterize_segments(
            xys, depths, radii, conics, num_tiles_hit, opac, h, w,
            [(feature, _zeros(d, dev)), (tail, bg_tail, (3, 1, 3))], **kw)
        return feat_im, rgb, depth, normal
    colors = torch.cat([feature, rgbs, depths[:, None], normals], dim=1)
    background = torch.zeros(d + 7, device=dev)
    background[d + 3] = 10.0
    img = ops.NDRasterizeGaussians.apply(xys, depths, radii, conics, num_tiles_hit, colors, opac, h, w,
                                         background)
    # one split (its backward is a single cat of the four cotangents, not four zero-padded adds)
    return torch.split(img, [d, 3, 1, 3], dim=-1)


def rasterize_activated_fused(act: Dict, view: ViewParams, ops, sh_degree_to_use: int = 4
                              ) -> Dict[str, torch.Tensor]:
    """the four outputs from ONE rasterize call — feature(D) | rgb(3) | depth(1) |
    normal(3) concatenated into one (N, D+7) colour tensor for NDRasterizeGaussians (the reference
    renders depth as 3 replicated channels and keeps one; here it is one channel).  Same images,
    bit for bit, as the four separate calls; one opacity activation, one set of geometry gradients.
    What a maintainer would call from `get_outputs` instead of :735-784."""
    dev = act["means"].device
    h, w = view.height, view.width
    opac = act["opac"]() if callable(act["opac"]) else act["opac"]
    xys, depths, radii, conics, num_tiles_hit, cov3d = ops.ProjectGaussians.apply(
        act["means"], act["scales"], 1, act["quats"], view.viewmat[:3, :].to(dev),
        view.projmat.to(dev), view.fx, view.fy, view.cx, view.cy, h, w, view.tile_bounds)
    if xys.requires_grad:
        xys.retain_grad()
    if hasattr(ops, "ShadeTail") and hasattr(ops, "rasterize_segments"):
        # SH + clamp + the (N, 7) rgb | depth | normal array in one kernel each way
        rgbs = None
        tail = ops.ShadeTail.apply(sh_degree_to_use, act["viewdirs"], act["sh"], depths, act["normals"])
    else:
        tail = None
        rgbs = ops.SphericalHarmonics.apply(sh_degree_to_use, act["viewdirs"], act["sh"])
        rgbs = torch.clamp(rgbs + 0.5, 0.0, 1.0)
    feature, rgb, depth, normal = fused_images(ops, xys, depths, radii, conics, num_tiles_hit, opac, h, w,
                                               act["feature"], rgbs, act["normals"], tail=tail)
    return {"xys": xys, "radii": radii, "depths": depths, "conics": conics,
            "num_tiles_hit": num_tiles_hit, "feature": feature, "rgb": rgb, "depth": depth,
            "normal": normal}


def render_view(scene: Scene, view: ViewParams, ops, sh_degree_to_use: int = 4,
                channels=CHANNELS, fused: bool = False) -> Dict[str, torch.Tensor]:
    if fused:      # the plugin route: fused activations, one rasterize operator
        return rasterize_activated_fused(activate_fused(scene, view, ops), view, ops, sh_degree_to_use)
    act = activate(scene, view, ops.quat_to_rotmat)
    return rasterize_activated(act, view, ops, sh_degree_to_use, channels)


def seeded_cotangents(outputs: Dict[str, torch.Tensor], seed: int = 0) -> Dict[str, torch.Tensor]:
    """Dense N(0,1) v_out for every image output, generated on the CPU so that CPU
    and GPU runs see the same numbers."""
    g = torch.Generator(device="cpu").manual_seed(seed)
    cot = {}
    for name in CHANNELS:
        if name in outputs:
            t = outputs[name]
            cot[name] = torch.randn(t.shape, generator=g, dtype=torch.float32).to(t.device)
    return cot


def backward_view(outputs: Dict[str, torch.Tensor], cotangents: Dict[str, torch.Tensor]) -> None:
    """One backward through all rasterize calls, SH and project with the given v_out tensors fed
    directly as the cotangents of the images (equivalent to loss = sum_k <output_k, v_out_k>, without
    spending device time on evaluating that loss)."""
    names = list(cotangents)
    torch.autograd.backward([outputs[k] for k in names], [cotangents[k] for k in names])
