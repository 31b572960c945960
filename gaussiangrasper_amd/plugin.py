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
               fx, fy, cx, cy, H, W, tile_bounds, sh_degree_to_use: int, ops=_ops, full_proj=None
               ) -> Optional[Dict[str, torch.Tensor]]:
    """Operator part of `get_outputs` (reference :699-784) with the four rasterize calls fused.
    Sets model.xys / model.radii / model.normals as the reference does.  None if nothing is visible.
    full_proj: `projmat @ viewmat` (:707) if the caller has it already (cached per dataset camera)."""
    if hasattr(ops, "ViewGeometry") and hasattr(ops, "rasterize_segments") and model.config.sh_degree > 0:
        # activations, projection, SH + clamp + the rgb | depth | normal array: ONE autograd node (ops.ViewGeometry: three
        # forward kernels; in the training step one backward kernel over the Gaussians instead of three)
        if full_proj is None:
            full_proj = projmat @ viewmat                                # :707
        model._gg_last_view = (viewmat, full_proj)
        model.xys, depths, model.radii, conics, num_tiles_hit, opac, tail, model.normals, packed = ops.ViewGeometry.apply(
            means, log_scales, quats, opacities, colors_all, cam_pos.reshape(-1)[:3], viewmat[:3, :], full_proj,
            fx, fy, cx, cy, H, W, tile_bounds, sh_degree_to_use)
        if model.training:
            model.xys.retain_grad()                                    # :724-725
        feat_im, rgb, depth_im, normal_im = fused_images(ops, model.xys, depths, model.radii, conics, num_tiles_hit,
                                                         opac, H, W, feature, None, model.normals, tail=tail,
                                                         packed=packed)
        count = ops.last_num_intersects(model.radii)     # None: the lists binned last are not this call's
        if count == 0 or (count is None and (model.radii).sum() == 0):   # :714, answered after the render (see below)
            return None
        return {"rgb": rgb, "feature": feat_im, "depth": depth_im, "normal": normal_im}
    fused_act = hasattr(ops, "ActivateGaussians")
    if fused_act:      # exp / normalise / sigmoid / view directions / normals: one kernel each way
        scales_e, quats_n, opac, viewdirs, model.normals = ops.ActivateGaussians.apply(
            means, log_scales, quats, opacities, cam_pos.reshape(-1)[:3])
    else:
        scales_e, quats_n = torch.exp(log_scales), quats / quats.norm(dim=-1, keepdim=True)
        opac = torch.sigmoid(opacities)
    if full_proj is None:
        full_proj = projmat @ viewmat                                # :707
    model._gg_last_view = (viewmat, full_proj)                        # (what the operators were given: tests, debugging)
    model.xys, depths, model.radii, conics, num_tiles_hit, _cov3d = ops.ProjectGaussians.apply(
        means, scales_e, 1, quats_n, viewmat[:3, :], full_proj, fx, fy, cx, cy, H, W, tile_bounds)
    # :714 `if (self.radii).sum() == 0: return background` is a host round trip in front of every render.  With the
    # product operators the answer comes for free AFTER the render: the tile lists' length was read back
    # asynchronously while the sort and the blend were being enqueued (ops.Binning.resolve), and it is 0 exactly
    # when no radius is positive (a positive radius covers at least one tile: csrc/project.hip).  Operator
    # namespaces without that read-back (the oracle-backed test operators) keep the reference's check.
    count_known_late = hasattr(ops, "last_num_intersects")
    if not count_known_late and (model.radii).sum() == 0:
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
    if count_known_late:
        count = ops.last_num_intersects(model.radii)     # None: the lists binned last are not this call's
        if count == 0 or (count is None and (model.radii).sum() == 0):
            return None
    return {"rgb": rgb, "feature": feat_im, "depth": depth_im, "normal": normal_im}


class LazyOutputs(dict):
    """The output dictionary of get_outputs with `normal_vis` / `feature_vis` computed when first read.
    The reference computes both on every call (:785-795) — a rank-3 `torch.pca_lowrank` of the (H W, 32) feature
    image included — although training reads neither (get_metrics_dict / get_loss_dict index rgb, depth, normal,
    feature: :804-935).  The lazy keys are NOT in the underlying dict storage until they are computed: `[]` reaches them
    through `__missing__`, and `__iter__` / `keys` are overridden, which makes CPython's `dict(out)`, `{**out}` and
    `other.update(out)` leave their C fast path and go through `keys()` + `[]` — a copy holds real tensors, never a
    placeholder."""

    def __init__(self, base: Dict, lazy: Dict):
        super().__init__(base)
        self._lazy = dict(lazy)

    def _force(self, key=None):
        for k in ([key] if key is not None else list(self._lazy)):
            fn = self._lazy.pop(k, None)
            if fn is not None:
                super().__setitem__(k, fn())

    def __missing__(self, key):
        if key in self._lazy:
            self._force(key)
            return super().__getitem__(key)
        raise KeyError(key)

    def __contains__(self, key):
        return super().__contains__(key) or key in self._lazy

    def __iter__(self):
        yield from super().__iter__()
        yield from list(self._lazy)

    def __len__(self):
        return super().__len__() + len(self._lazy)

    def keys(self):
        return list(iter(self))

    def get(self, key, default=None):
        return self[key] if key in self else default

    def setdefault(self, key, default=None):
        if key in self:
            return self[key]
        super().__setitem__(key, default)
        return default

    def pop(self, key, *a):
        self._force(key)
        return super().pop(key, *a)

    def items(self):
        self._force()
        return super().items()

    def values(self):
        self._force()
        return super().values()

    def copy(self):
        self._force()
        return dict(super().items())

    def __eq__(self, other):
        self._force()
        return dict.__eq__(self, other)

    __hash__ = None

    def __or__(self, other):
        return {**self.copy(), **other}

    def __ror__(self, other):
        return {**other, **self.copy()}

    def __reduce__(self):
        return (dict, (self.copy(),))

    def __setitem__(self, key, value):
        self._lazy.pop(key, None)
        super().__setitem__(key, value)

    def __delitem__(self, key):
        if self._lazy.pop(key, None) is None:
            super().__delitem__(key)


def _camera_scalars(camera):
    """fx, fy, cx, cy, the fovs' arguments, W, H of a one-camera `Cameras` in ONE device-to-host copy (the
    reference reads them with six `.item()` / implicit float() conversions, each a stream synchronisation when the
    camera lives on the GPU, as the datamanager's does: gaussian_splatting.py:669-674,706-707).  The two fov
    arguments are the fp32 quotients the reference forms on the device (`camera.width / (2 * camera.fx)`: int64 by
    float32 divides in float32), formed here from the same fp32 values."""
    import numpy as np
    vals = torch.cat([camera.fx.reshape(1), camera.fy.reshape(1), camera.cx.reshape(1), camera.cy.reshape(1),
                      camera.width.reshape(1), camera.height.reshape(1)]).tolist()      # (promoted to fp32: exact)
    fx, fy, cx, cy = vals[:4]
    w, h = int(round(vals[4])), int(round(vals[5]))
    f32 = np.float32
    ax = float(f32(w) / (f32(2.0) * f32(fx)))
    ay = float(f32(h) / (f32(2.0) * f32(fy)))
    return fx, fy, cx, cy, ax, ay, w, h


def _default_loss_ops():
    from . import losses
    return losses


def _default_mlp_class():
    from .mlp import MLP
    return MLP


def _resize_image(mod, img_hwc, newsize):
    """`TF.resize(img.permute(2, 0, 1), newsize, antialias=None).permute(1, 2, 0)` (gaussian_splatting.py:849-851) with
    the reference module's own torchvision if it has one; without torchvision, the call torchvision makes for a
    tensor with antialias off: bilinear interpolation, align_corners=False."""
    tf = getattr(mod, "TF", None)
    chw = img_hwc.permute(2, 0, 1)
    if tf is not None:
        return tf.resize(chw, newsize, antialias=None).permute(1, 2, 0)
    return torch.nn.functional.interpolate(chw[None], size=tuple(newsize), mode="bilinear", align_corners=False,
                                           antialias=False)[0].permute(1, 2, 0)


def _nearest_columns(src_chw: torch.Tensor, size, pixels: torch.Tensor) -> torch.Tensor:
    """`F.interpolate(src[None], size=size, mode="nearest")[0][:, pixels[:, 0], pixels[:, 1]]` without the (C, H, W)
    intermediate (512 x 1200 x 1600 floats = 3.9 GB per view at the bench size; the reference builds it at :875-876 and
    then reads <= 1000 columns of it, :918).  ATen's nearest rule: source index = min(floor(dst * float(in / out)), in - 1)."""
    c, hs, ws = src_chw.shape
    sy = torch.tensor(hs / size[0], dtype=torch.float32)
    sx = torch.tensor(ws / size[1], dtype=torch.float32)
    iy = torch.clamp((pixels[:, 0].to(torch.float32) * sy.to(pixels.device)).floor().long(), max=hs - 1)
    ix = torch.clamp((pixels[:, 1].to(torch.float32) * sx.to(pixels.device)).floor().long(), max=ws - 1)
    return src_chw[:, iy, ix]


def device_sampling_default() -> bool:
    """GG_DEVICE_SAMPLING=1: the feature losses' pixel samples come from gaussiangrasper_amd.sampling (device generator)
    instead of the reference's helpers (host `torch.randperm` over every label's pixel count: ~0.3 s per 1600x1200 view)."""
    import os
    return os.environ.get("GG_DEVICE_SAMPLING", "0") not in ("0", "false", "no", "")


def make_fused_model_class(base, ops=_ops, background_override=lambda: None, fused_training: bool = False,
                           loss_ops=None, mlp_class="default", device_sampling: Optional[bool] = None):
    """Subclass of the reference's GaussianSplattingModel whose get_outputs uses `fused_view`.
    `base` is nerfstudio.models.gaussian_splatting.GaussianSplattingModel (or `stub.StubGaussianSplattingModel`,
    which restates the attributes used here, where nerfstudio is not installed).
    fused_training: the optimizer-side callbacks run on csrc/densify.hip too (`after_train`,
    `refinement_after` through `densify.Refiner`: statistics, masks, split / duplicate / cull and the Adam-state
    surgery of the six groups in a handful of launches, reference :373-546), `get_loss_dict` (:841-935) computes its
    image-space terms with the fused loss kernels (`loss_ops`, default gaussiangrasper_amd.losses: main_loss,
    depth_normal_loss, cosine_similarity_loss, gather_pixels) and `fea_up` (:258) becomes `mlp_class` (default
    gaussiangrasper_amd.mlp.MLP: same sub-modules and state-dict keys, fused forward / backward kernels; None keeps the
    reference's module).
    device_sampling: draw the feature losses' pixel samples with gaussiangrasper_amd.sampling (same law, device
    generator, ~1 ms) instead of the reference module's helpers (same draws as the reference, ~0.3 s per full-size view
    of host `torch.randperm`); None: GG_DEVICE_SAMPLING (default off)."""
    if device_sampling is None:
        device_sampling = device_sampling_default()

    class FusedGaussianSplattingModel(base):
        """GaussianSplattingModel on the fused MI355X rasterizer call (gaussiangrasper_amd.plugin)."""

        # "lazy": normal_vis / feature_vis (the reference's rank-3 PCA of the feature image, :785-795) are computed when
        # something reads them — `[]`, `get`, `items`, `values`, iteration and every kind of copy see tensors (LazyOutputs);
        # "eager": on every call, in a plain dict, as the reference does; "off": feature_vis = first 3 channels
        feature_vis_mode = "lazy"

        def get_outputs(self, camera) -> Dict[str, Union[torch.Tensor, List]]:
            if not hasattr(camera, "camera_to_worlds"):            # :633-635
                print("Called get_outputs with not a camera")
                return {}
            assert camera.shape[0] == 1, "Only one camera at a time"
            if self.training:
                self.camera_optimizer.apply_to_camera(camera)
                # :642 `background = torch.rand(feature_dim)` is read by the two early exits only (:652 is eval-only, :715
                # "nothing visible"): drawn where it is returned, not on every call (a Philox launch per view; the
                # device RNG stream then differs from the reference's from the first step on: PARITY.md)
                background = None
            else:
                over = background_override()
                background = over if over is not None else self.back_color.to(self.device)
            crop_ids = None
            if self.crop_box is not None and not self.training:   # :649-652
                crop_ids = self.crop_box.within(self.means).squeeze()
                if crop_ids.sum() == 0:
                    return {"rgb": background.repeat(camera.height.item(), camera.width.item(), 1)}
            camera_downscale = self._get_downscale_factor()
            if camera_downscale != 1:     # (x 1.0 changes nothing: seven launches and a host-to-device copy saved)
                camera.rescale_output_resolution(1 / camera_downscale)
            md = getattr(camera, "metadata", None)
            key = (int(md["cam_idx"]), camera_downscale) if (self.training and isinstance(md, dict)
                                                            and "cam_idx" in md) else None
            # With the camera optimizer off (the reference's configuration: CameraOptimizerConfig(mode="off"), :191) a
            # training camera's pose is a dataset constant like its intrinsics: viewmat and projmat @ viewmat of a
            # (dataset index, downscale) are formed once — the same torch operations on the same values, i.e. the same
            # bits — instead of ~10 small launches per view (a 3x3 and two 4x4 products through hipBLASLt, eye, fills,
            # neg, copies: 12 % of a view's wall time in round 3's rocprofv3 table)
            pose_const = getattr(getattr(getattr(self, "config", None), "camera_optimizer", None), "mode", "off") == "off"
            vcache = self.__dict__.setdefault("_gg_view_cache", {})
            view_c = vcache.get(key) if (key is not None and pose_const) else None
            if view_c is None:
                # world -> camera, gsplat convention: rotate pi about x, analytic inverse (:658-668)
                c2w = camera.camera_to_worlds[0]
                flips = self.__dict__.setdefault("_gg_flip", {})
                fkey = (str(c2w.device), c2w.dtype)
                if fkey not in flips:
                    flips[fkey] = torch.diag(torch.tensor([1.0, -1.0, -1.0], device=c2w.device, dtype=c2w.dtype))
                R = c2w[:3, :3] @ flips[fkey]
                R_inv = R.T
                viewmat = torch.eye(4, device=c2w.device, dtype=c2w.dtype)
                viewmat[:3, :3] = R_inv
                viewmat[:3, 3:4] = -R_inv @ c2w[:3, 3:4]
            else:
                viewmat = view_c[0]
            # The intrinsics are host values in every operator signature, and reading them off a device-resident
            # camera is a stream synchronisation: the host then cannot enqueue view k + 1 while view k runs (measured
            # on the bench workload: 520 -> 459 views/s with one read-back per view).  The datamanager stamps every
            # training camera with its dataset index (full_images_datamanager.py:375-377, `metadata["cam_idx"]`)
            # and intrinsics are per-dataset constants (camera optimisation, if on, moves poses only), so each
            # (index, downscale) is read back once.  Cameras without the stamp, and eval mode (where train and eval
            # datasets share index values), are read once per set of camera TENSORS (address + version).
            cache = self.__dict__.setdefault("_gg_camera_scalars", {})
            scal = cache.get(key) if key is not None else None
            if scal is None:
                # (unstamped / eval cameras: the SAME tensors at the same versions hold the same values — a render loop
                #  that keeps its camera objects, as the viewer and bench.py --config 5 do, is read back once per object)
                tkey = tuple((t.data_ptr(), t._version, str(t.device)) for t in
                             (camera.fx, camera.fy, camera.cx, camera.cy, camera.width, camera.height))
                tcache = self.__dict__.setdefault("_gg_camera_scalars_by_tensor", {})
                hit = tcache.get(tkey)
                if hit is None:
                    if len(tcache) >= 256:
                        tcache.clear()
                    hit = tcache[tkey] = (_camera_scalars(camera), (camera.fx, camera.fy, camera.cx, camera.cy,
                                                                  camera.width, camera.height))   # (refs: addresses stay theirs)
                scal = hit[0]
                if key is not None:
                    cache[key] = scal
            fx, fy, cx, cy, ax, ay, W, H = scal
            fovx, fovy = 2 * math.atan(ax), 2 * math.atan(ay)       # :672-673 (fp32 quotients, atan in double)
            self.last_size = (H, W)
            pcache = self.__dict__.setdefault("_gg_projmat", {})
            pkey = (ax, ay, str(self.device))
            if pkey not in pcache:       # a function of the intrinsics only: one host-built matrix per camera model
                pcache[pkey] = projection_matrix(0.001, 1000, fovx, fovy, device=self.device)
            projmat = pcache[pkey]
            tile_bounds = ((W + BLOCK - 1) // BLOCK, (H + BLOCK - 1) // BLOCK, 1)
            pick = (lambda t: t[crop_ids]) if crop_ids is not None else (lambda t: t)
            if view_c is None:
                cam_pos = camera.camera_to_worlds.detach()[..., :3, 3]
                full_proj = projmat @ viewmat                        # :707
                if key is not None and pose_const:
                    vcache[key] = (viewmat, full_proj, cam_pos.clone())
            else:
                _, full_proj, cam_pos = view_c
            n = min(self.step // self.config.sh_degree_interval, self.config.sh_degree)
            out = fused_view(self, pick(self.means), pick(self.scales), pick(self.quats), pick(self.opacities),
                             pick(self.colors_all), pick(self.feature), viewmat, projmat, cam_pos, fx, fy, cx, cy,
                             H, W, tile_bounds, n, ops, full_proj=full_proj)
            if out is None:
                # :714-715 — the reference leaves through this exit WITHOUT scaling the camera back (:798 is not
                # reached); kept as it is (PARITY.md)
                if background is None:
                    background = torch.rand(self.feature_dim, device=self.device)     # :642
                return {"rgb": background.repeat(H, W, 1)}
            if camera_downscale != 1:
                camera.rescale_output_resolution(camera_downscale)  # :798
            feat, normal = out["feature"], out["normal"]

            def normal_vis():
                return (normal + 1) / 2

            def feature_vis():
                if feat.shape[-1] == 3:
                    return (torch.nn.functional.normalize(feat, dim=-1) + 1) / 2
                if self.feature_vis_mode == "off":
                    return feat[..., :3]
                flat = feat.view(-1, feat.size(-1))
                _, _, V = torch.pca_lowrank(flat, q=3)
                return torch.matmul(flat, V[:, :3]).view(feat.size()[:-1] + (3,))

            lazy = {"normal_vis": normal_vis, "feature_vis": feature_vis}
            if self.feature_vis_mode == "lazy":
                return LazyOutputs(out, lazy)
            out.update({k: f() for k, f in lazy.items()})
            return out

        # ---- losses and fea_up (fused_training) --------------------------------------------------
        def populate_modules(self):
            super().populate_modules()
            cls = _default_mlp_class() if mlp_class == "default" else mlp_class
            if fused_training and cls is not None and hasattr(self, "fea_up"):
                old = self.fea_up
                l0, l2 = old.layers[0], old.layers[-1]
                new = cls(l0.in_features, l2.out_features, hidden_list=[l0.out_features])
                new.load_state_dict(old.state_dict())        # same keys: fea_up.layers.{0,2}.{weight,bias}
                self.fea_up = new.to(l0.weight.device)

        def get_loss_dict(self, outputs, batch, metrics_dict=None) -> Dict[str, torch.Tensor]:
            """gaussian_splatting.py:841-935 with the same keys, the same ground-truth preparation (resize /
            interpolate / masks, :846-876), the same sampling helpers and random draws (:909-910) and the same
            regularisers (:920-929); the image-space terms run on the fused kernels:
              main_loss (:882-885, :931)      one pass each way instead of ten grouped conv2d launches each way;
              depth_loss, normal_loss (:879-880)  one pass each way, no boolean-index gathers;
              feature_loss, up_loss (:911-918)    ONE gather of all sampled pixel sets, fused cosine losses, `fea_up`
                                               on the MLP kernels; the nearest-neighbour up-sampling of the
                                               (h, w, 512) ground-truth feature map is evaluated at the <= 1000 sampled
                                               pixels only (same values, no (512, H, W) intermediate).
            Unlike the reference it does not zero `gt_img` / `outputs["rgb"]` at the invalid pixels in place (:883-884:
            a side effect of how the reference feeds its SSIM; nothing reads either afterwards)."""
            if not fused_training:
                return super().get_loss_dict(outputs, batch, metrics_dict)
            import sys
            F = torch.nn.functional
            L = loss_ops if loss_ops is not None else _default_loss_ops()
            mod = sys.modules.get(base.__module__)
            if device_sampling:
                from .sampling import sampling_in_mask, sampling_pairs_in_mask
            else:
                sampling_pairs_in_mask, sampling_in_mask = mod.sampling_pairs_in_mask, mod.sampling_in_mask
            d = self._get_downscale_factor()
            if d > 1:
                newsize = [batch["image"].shape[0] // d, batch["image"].shape[1] // d]
                gt_img = _resize_image(mod, batch["image"], newsize)
            else:
                gt_img = batch["image"]
            size = (gt_img.shape[0], gt_img.shape[1])
            gt_normal = batch["normal"].permute(2, 0, 1).unsqueeze(0).to(self.device)
            gt_normal = F.interpolate(gt_normal, size=size, mode='bilinear').squeeze(0)
            gt_normal = F.normalize(gt_normal, dim=0)
            gt_depth = batch["depth"].permute(2, 0, 1).unsqueeze(0).to(self.device)
            depth_mask = (gt_depth > 0.05) * 1.0
            gt_depth = F.interpolate(gt_depth, size=size, mode='bilinear').squeeze(0)
            depth_mask = F.interpolate(depth_mask, size=size, mode='nearest').squeeze(0)
            gt_mask = batch["sam_mask"].to(self.device)
            gt_mask = F.interpolate(gt_mask.float().unsqueeze(0).unsqueeze(0), size=size,
                                    mode='nearest').squeeze(0).squeeze(0)
            valid_mask = batch["valid_mask"].to(self.device)
            valid_mask = F.interpolate(valid_mask.float().unsqueeze(0).unsqueeze(0), size=size,
                                       mode='nearest').squeeze(0).squeeze(0)
            depth_mask = depth_mask * valid_mask
            depth_mask = depth_mask > 0
            valid_mask = valid_mask > 0
            gt_mask[~valid_mask] = -1.0
            gt_fea = batch["feature"].permute(2, 0, 1).float().to(self.device)      # (:875-876: read below, at the samples)

            depth_loss, normal_loss = L.depth_normal_loss(outputs["depth"], gt_depth, outputs["normal"], gt_normal,
                                                          depth_mask[0])                       # :879-880
            main_loss = L.main_loss(outputs["rgb"], gt_img.to(self.device), valid_mask,
                                    self.config.ssim_lambda)[0]                                 # :882-885, :931

            feature = outputs["feature"]
            selected_pairs = sampling_pairs_in_mask(gt_mask, 800)                               # :909
            selected_points = sampling_in_mask(gt_mask, 1000)                                   # :910
            sets = [p for pair in selected_pairs for p in pair] + [selected_points]
            rows = L.gather_pixels(feature, *sets)             # the reference's 2 len(pairs) + 1 gathers (:912-917) in one
            fea_loss = 0
            for i in range(len(selected_pairs)):
                fea_loss += L.cosine_similarity_loss(rows[2 * i].permute(1, 0), rows[2 * i + 1].permute(1, 0))
            fea_loss = fea_loss / len(selected_pairs)
            fea_up = self.fea_up(rows[-1]).permute(1, 0)
            up_loss = L.cosine_similarity_loss(fea_up, _nearest_columns(gt_fea, size, selected_points))

            if self.step % 10 == 0:                                                             # :920-929, as they are
                sh_reg = self.colors_all[:, 1:, :].norm(dim=1).mean()
                scale_exp = torch.exp(self.scales)
                scale_reg = torch.maximum(scale_exp.amax(dim=-1) / scale_exp.amin(dim=-1),
                                          torch.tensor(self.config.max_gauss_ratio)) - self.config.max_gauss_ratio
                scale_reg = 0.1 * scale_reg.mean()
            else:
                sh_reg = torch.tensor(0.0).to(self.device)
                scale_reg = torch.tensor(0.0).to(self.device)
            return {"main_loss": main_loss, "feature_loss": fea_loss, "up_loss": up_loss, "depth_loss": depth_loss,
                    "normal_loss": normal_loss, "sh_reg": sh_reg, "scale_reg": scale_reg}

        # ---- optimizer side (fused_training) -----------------------------------------------------
        def _refiner(self, optimizers=None):
            from .densify import GROUPS, RefineConfig, Refiner
            r = getattr(self, "_gg_refiner", None)
            if r is None:
                c = self.config
                cfg = RefineConfig(**{k: getattr(c, k) for k in RefineConfig.__dataclass_fields__ if hasattr(c, k)})
                r = Refiner({a: getattr(self, a) for a in GROUPS.values()}, {}, cfg,
                            num_train_data=getattr(self, "num_train_data", 0))
                object.__setattr__(self, "_gg_refiner", r)
            if optimizers is not None:   # nerfstudio's Optimizers: .optimizers maps group name -> torch optimizer
                r.optimizers = {g: o for g, o in optimizers.optimizers.items() if g in GROUPS}
            return r

        def after_train(self, step: int):
            if not fused_training:
                return super().after_train(step)
            assert step == self.step
            if self.xys.grad is None:
                raise RuntimeError("after_train: self.xys has no gradient (get_outputs retains it in training mode)")
            r = self._refiner()
            r.params = {a: getattr(self, a) for a in r.params}      # (load_state_dict may have replaced them)
            r.after_train(self.xys.grad, self.radii, self.last_size)
            self.xys_grad_norm, self.vis_counts, self.max_2Dsize = r.xys_grad_norm, r.vis_counts, r.max_2Dsize

        def refinement_after(self, optimizers, step):
            if not fused_training:
                return super().refinement_after(optimizers, step)
            assert step == self.step
            r = self._refiner(optimizers)
            r.params = {a: getattr(self, a) for a in r.params}
            info = r.refinement_after(step)
            from .densify import GROUPS
            for group, attr in GROUPS.items():
                setattr(self, attr, r.params[attr])                 # the new Parameters (:434-439, :497-502)
                if hasattr(optimizers, "parameters") and group in optimizers.parameters:
                    optimizers.parameters[group] = [r.params[attr]]
            self.xys_grad_norm, self.vis_counts, self.max_2Dsize = r.xys_grad_norm, r.vis_counts, r.max_2Dsize
            return info

    FusedGaussianSplattingModel.__qualname__ = "FusedGaussianSplattingModel"
    return FusedGaussianSplattingModel


_model_class = {}
FUSED_GROUPS = ("xyz", "color", "feature", "opacity", "scaling", "rotation")   # method_configs.py:618-660


def fused_training_default() -> bool:
    """GG_FUSED_TRAINING=0 keeps the reference's torch.optim.Adam instances and its own refinement callbacks."""
    import os
    return os.environ.get("GG_FUSED_TRAINING", "1") not in ("0", "false", "no", "")


def model_class(fused_training: bool = False):
    """The subclass of the real reference model (imports nerfstudio)."""
    if fused_training not in _model_class:
        from nerfstudio.model_components import renderers
        from nerfstudio.models.gaussian_splatting import GaussianSplattingModel
        _model_class[fused_training] = make_fused_model_class(
            GaussianSplattingModel, background_override=lambda: renderers.BACKGROUND_COLOR_OVERRIDE,
            fused_training=fused_training)
    return _model_class[fused_training]


def _spec(method_name: str, fused_training: Optional[bool] = None):
    from nerfstudio.configs.method_configs import method_configs    # populated before plugins are discovered
    from nerfstudio.plugins.types import MethodSpecification
    if fused_training is None:
        fused_training = fused_training_default()
    config = copy.deepcopy(method_configs["gaussian-splatting"])
    config.method_name = method_name
    config.pipeline.model._target = model_class(fused_training)
    if fused_training:
        # engine/optimizers.py:45-58: AdamOptimizerConfig.setup(params) calls _target(params, lr=..., eps=..., ...):
        # the six Gaussian groups step through ONE streaming kernel each (optim.FusedAdam: torch.optim.Adam's
        # constructor, param_groups and state, bit-identical update); camera_opt / up_net stay torch.optim.Adam
        from .optim import FusedAdam
        for group in FUSED_GROUPS:
            if group in config.optimizers:
                config.optimizers[group]["optimizer"]._target = FusedAdam
    return MethodSpecification(config=config, description=DESCRIPTION)


def gaussian_splatting():
    """NERFSTUDIO_METHOD_CONFIGS="gaussian-splatting=gaussiangrasper_amd.plugin:gaussian_splatting":
    replaces the reference's method of that name, so train.sh / render.sh / update.sh run unchanged — the fused
    renderer, and (unless GG_FUSED_TRAINING=0) the fused Adam step and refinement callbacks."""
    return _spec("gaussian-splatting")


def gaussian_splatting_amd():
    """NERFSTUDIO_METHOD_CONFIGS="gaussian-splatting-amd=gaussiangrasper_amd.plugin:gaussian_splatting_amd":
    the same method next to the reference's."""
    return _spec("gaussian-splatting-amd")


def __getattr__(name: str):
    """Entry-point route.  `discover_methods` (nerfstudio/plugins/registry.py:42-51) takes what an entry point of
    group `nerfstudio.method_configs` loads only if it already IS a MethodSpecification (callables are called for
    the environment variable only, :65-66), so the module offers the two specifications as attributes built on
    first access (PEP 562):

        [project.entry-points."nerfstudio.method_configs"]
        gaussian-splatting-amd = "gaussiangrasper_amd.plugin:gaussian_splatting_amd_spec"
    """
    if name == "gaussian_splatting_spec":
        return gaussian_splatting()
    if name == "gaussian_splatting_amd_spec":
        return gaussian_splatting_amd()
    raise AttributeError(f"module {__name__!r} has no attribute {name!r}")
