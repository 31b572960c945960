"""nerfstudio method plugin: the fused rasterizer path behind the reference's UNCHANGED entry points.

The reference discovers methods through the entry-point group `nerfstudio.method_configs` or the
environment variable NERFSTUDIO_METHOD_CONFIGS="name=module:attr" (nerfstudio/plugins/registry.py:34-79),
takes a `MethodSpecification(config, description)` (nerfstudio/plugins/types.py:23-33) and merges what
it finds into its method table with overwrite=True (nerfstudio/configs/method_configs.py:668-693).  So

    NERFSTUDIO_METHOD_CONFIGS="gaussian-splatting=gaussiangrasper_amd.plugin:gaussian_splatting" \\
    PYTHONPATH=<repo>/shim:<repo> ./train.sh            # or render.sh / update.sh, unchanged

runs `ns-train gaussian-splatting` with the reference's own TrainerConfig — same datamanager, optimizers,
schedules, losses, callbacks — and only `pipeline.model._target` replaced by the subclass below, whose
`get_outputs` renders the four images of one view from ONE fused rasterize call (feature | rgb | depth |
normal as a 39-channel colour tensor: one binning, one forward walk, one backward walk and one set of
geometry gradients instead of four; bit-identical images) and keeps every side effect the
rest of the model relies on (`self.xys` with retain_grad, `self.radii`, `self.last_size`,
`self.normals`; reference gaussian_splatting.py:624-802).  `gaussian-splatting-amd` registers the same
method under its own name next to the reference's.

The module imports nerfstudio lazily: `make_fused_model_class(Base)` only needs the base class (the tests
drive it with a stub), `gaussian_splatting()` builds the MethodSpecification on demand."""
from __future__ import annotations

import copy
import math
from typing import Dict, List, Optional, Union

import torch

from . import ops as _ops
from .camera import projection_matrix
from .constants import BLOCK
from .pipeline import fused_images

DESCRIPTION = ("GaussianGrasper feature-field splatting on the MI355X-native fused rasterizer "
               "(gaussiangrasper_amd): one rasterize call per view for rgb + feature + depth + normal")


def fused_view(model, means, log_scales, quats, opacities, colors_all, feature, viewmat, projmat, cam_pos,
               fx, fy, cx, cy, H, W, tile_bounds, sh_degree_to_use: int, ops=_ops
               ) -> Optional[Dict[str, torch.Tensor]]:
    """Operator part of `get_outputs` (reference :699-784) with the four rasterize calls fused.
    Sets model.xys / model.radii / model.normals as the reference does.  None if nothing is visible."""
    fused_act = hasattr(ops, "ActivateGaussians")
    if fused_act:      # exp / normalise / sigmoid / view directions / normals: one kernel each way
        scales_e, quats_n, opac, viewdirs, model.normals = ops.ActivateGaussians.apply(
            means, log_scales, quats, opacities, cam_pos.reshape(-1)[:3])
    else:
        scales_e, quats_n = torch.exp(log_scales), quats / quats.norm(dim=-1, keepdim=True)
        opac = torch.sigmoid(opacities)
    model.xys, depths, model.radii, conics, num_tiles_hit, _cov3d = ops.ProjectGaussians.apply(
        means, scales_e, 1, quats_n, viewmat[:3, :], projmat @ viewmat, fx, fy, cx, cy, H, W, tile_bounds)
    if (model.radii).sum() == 0:                                   # :714
        return None
    if model.training:
        model.xys.retain_grad()                                    # :724-725
    if not fused_act:  # smallest-axis normals (:605-619) of the rendered subset
        rot = ops.quat_to_rotmat(quats)
        idx = log_scales.exp().min(dim=-1)[1][..., None, None].expand(-1, 3, -1)
        model.normals = rot.gather(2, idx).squeeze(dim=2)
    tail = None
    if model.config.sh_degree > 0:
        if not fused_act:
            viewdirs = means.detach() - cam_pos                    # :727-728
            viewdirs = viewdirs / viewdirs.norm(dim=-1, keepdim=True)
        if hasattr(ops, "ShadeTail") and hasattr(ops, "rasterize_segments"):
            # SH, + 0.5, clamp (:730-731) and the rgb | depth | normal colour array in one kernel each way
            rgbs = None
            tail = ops.ShadeTail.apply(sh_degree_to_use, viewdirs, colors_all, depths, model.normals)
        else:
            rgbs = ops.SphericalHarmonics.apply(sh_degree_to_use, viewdirs, colors_all)
            rgbs = torch.clamp(rgbs + 0.5, 0.0, 1.0)               # :731
    else:
        rgbs = torch.sigmoid(colors_all[:, 0, :])
    # feature | rgb | depth (background 10, :769) | normal from one binning (pipeline.fused_images)
    feat_im, rgb, depth_im, normal_im = fused_images(ops, model.xys, depths, model.radii, conics, num_tiles_hit,
                                                     opac, H, W, feature, rgbs, model.normals, tail=tail)
    return {"rgb": rgb, "feature": feat_im, "depth": depth_im, "normal": normal_im}


def make_fused_model_class(base, ops=_ops, background_override=lambda: None):
    """Subclass of the reference's GaussianSplattingModel whose get_outputs uses `fused_view`.
    `base` is nerfstudio.models.gaussian_splatting.GaussianSplattingModel (or a stub with the same
    attributes in the tests)."""

    class FusedGaussianSplattingModel(base):
        """GaussianSplattingModel on the fused MI355X rasterizer call (gaussiangrasper_amd.plugin)."""

        compute_feature_vis = True     # the reference runs a rank-3 PCA of the feature image every call (:792-795)

        def get_outputs(self, camera) -> Dict[str, Union[torch.Tensor, List]]:
            if not hasattr(camera, "camera_to_worlds"):            # :633-635
                print("Called get_outputs with not a camera")
                return {}
            assert camera.shape[0] == 1, "Only one camera at a time"
            if self.training:
                self.camera_optimizer.apply_to_camera(camera)
                background = torch.rand(self.feature_dim, device=self.device)
            else:
                over = background_override()
                background = over if over is not None else self.back_color.to(self.device)
            crop_ids = None
            if self.crop_box is not None and not self.training:   # :649-652
                crop_ids = self.crop_box.within(self.means).squeeze()
                if crop_ids.sum() == 0:
                    return {"rgb": background.repeat(camera.height.item(), camera.width.item(), 1)}
            camera_downscale = self._get_downscale_factor()
            camera.rescale_output_resolution(1 / camera_downscale)
            # world -> camera, gsplat convention: rotate pi about x, analytic inverse (:658-668)
            c2w = camera.camera_to_worlds[0]
            R = c2w[:3, :3] @ torch.diag(torch.tensor([1.0, -1.0, -1.0], device=c2w.device, dtype=c2w.dtype))
            R_inv = R.T
            viewmat = torch.eye(4, device=c2w.device, dtype=c2w.dtype)
            viewmat[:3, :3] = R_inv
            viewmat[:3, 3:4] = -R_inv @ c2w[:3, 3:4]
            cx, cy = camera.cx.item(), camera.cy.item()
            fx, fy = camera.fx.item(), camera.fy.item()
            W, H = camera.width.item(), camera.height.item()
            fovx, fovy = 2 * math.atan(W / (2 * fx)), 2 * math.atan(H / (2 * fy))
            self.last_size = (H, W)
            projmat = projection_matrix(0.001, 1000, fovx, fovy, device=self.device)
            tile_bounds = ((W + BLOCK - 1) // BLOCK, (H + BLOCK - 1) // BLOCK, 1)
            pick = (lambda t: t[crop_ids]) if crop_ids is not None else (lambda t: t)
            cam_pos = camera.camera_to_worlds.detach()[..., :3, 3]
            n = min(self.step // self.config.sh_degree_interval, self.config.sh_degree)
            out = fused_view(self, pick(self.means), pick(self.scales), pick(self.quats), pick(self.opacities),
                             pick(self.colors_all), pick(self.feature), viewmat, projmat, cam_pos, fx, fy, cx, cy,
                             H, W, tile_bounds, n, ops)
            camera.rescale_output_resolution(camera_downscale)      # :798 (both exits)
            if out is None:
                return {"rgb": background.repeat(camera.height.item(), camera.width.item(), 1)}
            out["normal_vis"] = (out["normal"] + 1) / 2
            feat = out["feature"]
            if feat.shape[-1] == 3:
                out["feature_vis"] = (torch.nn.functional.normalize(feat, dim=-1) + 1) / 2
            elif self.compute_feature_vis:
                flat = feat.view(-1, feat.size(-1))
                _, _, V = torch.pca_lowrank(flat, q=3)
                out["feature_vis"] = torch.matmul(flat, V[:, :3]).view(feat.size()[:-1] + (3,))
            else:
                out["feature_vis"] = feat[..., :3]
            return out

    FusedGaussianSplattingModel.__qualname__ = "FusedGaussianSplattingModel"
    return FusedGaussianSplattingModel


_model_class = None


def model_class():
    """The subclass of the real reference model (imports nerfstudio)."""
    global _model_class
    if _model_class is None:
        from nerfstudio.model_components import renderers
        from nerfstudio.models.gaussian_splatting import GaussianSplattingModel
        _model_class = make_fused_model_class(GaussianSplattingModel,
                                              background_override=lambda: renderers.BACKGROUND_COLOR_OVERRIDE)
    return _model_class


def _spec(method_name: str):
    from nerfstudio.configs.method_configs import method_configs    # populated before plugins are discovered
    from nerfstudio.plugins.types import MethodSpecification
    config = copy.deepcopy(method_configs["gaussian-splatting"])
    config.method_name = method_name
    config.pipeline.model._target = model_class()
    return MethodSpecification(config=config, description=DESCRIPTION)


def gaussian_splatting():
    """NERFSTUDIO_METHOD_CONFIGS="gaussian-splatting=gaussiangrasper_amd.plugin:gaussian_splatting":
    replaces the reference's method of that name, so train.sh / render.sh / update.sh run unchanged."""
    return _spec("gaussian-splatting")


def gaussian_splatting_amd():
    """NERFSTUDIO_METHOD_CONFIGS="gaussian-splatting-amd=gaussiangrasper_amd.plugin:gaussian_splatting_amd"
    (or the entry point of the same name): the fused method next to the reference's."""
    return _spec("gaussian-splatting-amd")
