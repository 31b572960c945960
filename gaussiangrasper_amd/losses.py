"""Losses of the reference's get_loss_dict on the HIP kernels: the cosine-similarity loss of the feature field
(below) and the image-space main loss (L1 + SSIM, `main_loss` at the end of the file).

Cosine-similarity loss: host-side mirror of the reference's
`cosine_similarity_loss(embeddings1, embeddings2)` (nerfstudio/models/gaussian_splatting.py:113-118):

    e1 = F.normalize(embeddings1, dim=0); e2 = F.normalize(embeddings2, dim=0)
    loss = 1 - (e1 * e2).sum(dim=0).mean()

with embeddings of shape (C, M) — the reference builds them as `.permute(1, 0)` of (M, C) gathers of the
rendered feature image (:909-918), so the (M, C) layout the kernels want is the storage they already
have.  Forward and backward are one launch each (`gg_cosine_loss_fwd/bwd`, csrc/losses.hip) instead of
~10 elementwise / reduction launches each way.  No CPU path."""
from __future__ import annotations

import torch
from torch import Tensor
from torch.autograd import Function

from . import _lib
from .ops import _f32, _ptr, _require_hip, _stream


class _CosineLoss(Function):
    @staticmethod
    def forward(ctx, a: Tensor, b: Tensor) -> Tensor:     # a, b: (M, C)
        dev = _require_hip(a, b)
        if a.shape != b.shape or a.dim() != 2:
            raise ValueError(f"embeddings must have the same 2-D shape, got {tuple(a.shape)} and {tuple(b.shape)}")
        a2, b2 = _f32(a), _f32(b)
        m, c = a2.shape
        sim = torch.empty(m, dtype=torch.float32, device=dev)
        na, nb = torch.empty_like(sim), torch.empty_like(sim)
        total = torch.empty(1, dtype=torch.float32, device=dev)
        _lib.check(_lib.load().gg_cosine_loss_fwd(m, c, _ptr(a2), _ptr(b2), _ptr(sim), _ptr(na), _ptr(nb),
                                                  _ptr(total), _stream(dev)), "gg_cosine_loss_fwd")
        ctx.save_for_backward(a2, b2, sim, na, nb)
        return 1.0 - total[0] / max(m, 1)

    @staticmethod
    def backward(ctx, v_loss: Tensor):
        a2, b2, sim, na, nb = ctx.saved_tensors
        dev = a2.device
        m, c = a2.shape
        v_a, v_b = torch.empty_like(a2), torch.empty_like(b2)
        vl = _f32(v_loss).reshape(1)
        _lib.check(_lib.load().gg_cosine_loss_bwd(m, c, _ptr(a2), _ptr(b2), _ptr(sim), _ptr(na), _ptr(nb),
                                                  _ptr(vl), _ptr(v_a), _ptr(v_b), _stream(dev)),
                   "gg_cosine_loss_bwd")
        return v_a, v_b


def cosine_similarity_loss(embeddings1: Tensor, embeddings2: Tensor) -> Tensor:
    """Drop-in for the reference function: embeddings (C, M), normalised along dim 0."""
    return _CosineLoss.apply(embeddings1.permute(1, 0), embeddings2.permute(1, 0))


# ------------------------------------------------------------------------------------------------
# image-space main loss: (1 - lambda) L1 + lambda (1 - SSIM)   (SURVEY.md §8f-4 tail)
# ------------------------------------------------------------------------------------------------
class _MainLoss(Function):
    @staticmethod
    def forward(ctx, rgb: Tensor, gt: Tensor, valid, ssim_lambda: float):
        dev = _require_hip(rgb, gt)
        if rgb.dim() != 3 or rgb.shape[2] != 3 or gt.shape != rgb.shape:
            raise ValueError(f"rgb and gt must be (H, W, 3), got {tuple(rgb.shape)} and {tuple(gt.shape)}")
        h, w = rgb.shape[:2]
        if h < 11 or w < 11:
            raise ValueError("image smaller than the 11 x 11 SSIM window")
        # the plugin route's rgb image is a channel slice of the (H, W, 7) image: read in place
        if rgb.dtype == torch.float32 and rgb.stride(2) == 1 and rgb.stride(1) >= 3 and rgb.stride(0) == w * rgb.stride(1):
            r = rgb
        else:
            r = _f32(rgb)
        rs = r.stride(1)
        g = _f32(gt)
        if valid is not None:
            if tuple(valid.shape) != (h, w):
                raise ValueError("valid_mask must be (H, W)")
            valid = valid.to(device=dev, dtype=torch.uint8).contiguous()
        lib = _lib.load()
        ws = torch.empty(lib.gg_image_loss_workspace(h, w), dtype=torch.uint8, device=dev)
        out3 = torch.empty(3, dtype=torch.float32, device=dev)
        _lib.check(lib.gg_image_loss_fwd(h, w, _ptr(r), rs, _ptr(g), _ptr(valid) if valid is not None else None,
                                         float(ssim_lambda), _ptr(out3), _ptr(ws), ws.numel(), _stream(dev)),
                   "gg_image_loss_fwd")
        ctx.save_for_backward(r, g, ws) if valid is None else ctx.save_for_backward(r, g, ws, valid)
        ctx.ssim_lambda = float(ssim_lambda)
        main, l1, ssim = out3[0], out3[1], out3[2]
        ctx.mark_non_differentiable(l1, ssim)
        return main, l1, ssim

    @staticmethod
    def backward(ctx, v_main, _v_l1, _v_ssim):
        saved = ctx.saved_tensors
        r, g, ws = saved[:3]
        valid = saved[3] if len(saved) > 3 else None
        dev = r.device
        h, w = r.shape[:2]
        v_rgb = torch.empty(h, w, 3, dtype=torch.float32, device=dev)
        vm = _f32(v_main).reshape(1)
        _lib.check(_lib.load().gg_image_loss_bwd(h, w, _ptr(r), r.stride(1), _ptr(g),
                                                 _ptr(valid) if valid is not None else None,
                                                 ctx.ssim_lambda, _ptr(vm), _ptr(ws), ws.numel(), _ptr(v_rgb),
                                                 _stream(dev)), "gg_image_loss_bwd")
        return v_rgb, None, None, None


def main_loss(rgb: Tensor, gt_img: Tensor, valid_mask=None, ssim_lambda: float = 0.2):
    """The reference's `main_loss` (nerfstudio/models/gaussian_splatting.py:882-885, :931) in one forward and one
    backward pass: returns (main_loss, Ll1, ssim) with
        Ll1 = |gt_img[valid_mask] - rgb[valid_mask]|.mean(),  ssim = SSIM(gt_img, rgb) on the images zeroed where
        invalid,  main_loss = (1 - ssim_lambda) Ll1 + ssim_lambda (1 - ssim);
    only main_loss carries a gradient (to rgb).  Unlike the reference it does not modify `gt_img` / `rgb` in place
    (the reference zeroes both at the invalid pixels as a side effect of computing the loss)."""
    return _MainLoss.apply(rgb, gt_img, valid_mask, ssim_lambda)


# ------------------------------------------------------------------------------------------------
# depth and normal losses over the masked pixels   (SURVEY.md §8f-4 tail; reference :879-880)
# ------------------------------------------------------------------------------------------------
def _pixel_layout(t: Tensor, h: int, w: int, channels: int):
    """(tensor, pixel stride, channel stride) in elements for an image given pixel-major (H, W, C) — possibly a
    channel slice of a wider image — or channel-major (C, H, W); anything else is made contiguous pixel-major."""
    if t.dtype == torch.float32:
        if t.dim() == 3 and tuple(t.shape) == (h, w, channels) and t.stride(0) == w * t.stride(1):
            return t, t.stride(1), t.stride(2) if channels > 1 else 1
        if t.dim() == 3 and tuple(t.shape) == (channels, h, w) and t.stride(1) == w * t.stride(2) and (h, w) != (w, channels):
            return t, t.stride(2), t.stride(0)
        if channels == 1 and t.dim() == 2 and tuple(t.shape) == (h, w) and t.stride(0) == w * t.stride(1):
            return t, t.stride(1), 1
    if t.dim() == 3 and tuple(t.shape) == (channels, h, w) and (h, w) != (w, channels):
        t = t.permute(1, 2, 0)
    t = _f32(t).reshape(h, w, channels)
    return t, channels, 1


class _DepthNormalLoss(Function):
    @staticmethod
    def forward(ctx, depth, gt_depth, normal, gt_normal, mask, h, w):
        dev = _require_hip(depth, gt_depth, normal, gt_normal)
        d, ds, _ = _pixel_layout(depth, h, w, 1)
        gd, gds, _ = _pixel_layout(gt_depth, h, w, 1)
        n, nps, ncs = _pixel_layout(normal, h, w, 3)
        gn, gps, gcs = _pixel_layout(gt_normal, h, w, 3)
        m = None if mask is None else mask.reshape(h, w).to(device=dev, dtype=torch.uint8).contiguous()
        lib = _lib.load()
        ws = torch.empty(lib.gg_geom_loss_workspace(), dtype=torch.uint8, device=dev)
        out3 = torch.empty(3, dtype=torch.float32, device=dev)
        _lib.check(lib.gg_geom_loss_fwd(h * w, _ptr(d), ds, _ptr(gd), gds, _ptr(n), nps, ncs, _ptr(gn), gps, gcs,
                                        _ptr(m) if m is not None else None, _ptr(out3), _ptr(ws), ws.numel(),
                                        _stream(dev)), "gg_geom_loss_fwd")
        ctx.layout = (h, w, ds, gds, nps, ncs, gps, gcs, tuple(depth.shape))
        ctx.has_mask = m is not None
        ctx.save_for_backward(*([d, gd, n, gn, ws] + ([m] if m is not None else [])))
        return out3[0], out3[1]

    @staticmethod
    def backward(ctx, v_dl, v_nl):
        saved = ctx.saved_tensors
        d, gd, n, gn, ws = saved[:5]
        m = saved[5] if ctx.has_mask else None
        h, w, ds, gds, nps, ncs, gps, gcs, dshape = ctx.layout
        dev = d.device
        z = lambda v: torch.zeros(1, dtype=torch.float32, device=dev) if v is None else _f32(v).reshape(1)
        v_depth = torch.empty(h * w, dtype=torch.float32, device=dev)
        v_normal = torch.empty(h, w, 3, dtype=torch.float32, device=dev)
        vd, vn = z(v_dl), z(v_nl)
        _lib.check(_lib.load().gg_geom_loss_bwd(h * w, _ptr(d), ds, _ptr(gd), gds, _ptr(n), nps, ncs, _ptr(gn), gps,
                                                gcs, _ptr(m) if m is not None else None, _ptr(vd), _ptr(vn),
                                                _ptr(ws), ws.numel(), _ptr(v_depth), _ptr(v_normal), _stream(dev)),
                   "gg_geom_loss_bwd")
        return v_depth.reshape(dshape), None, v_normal, None, None, None, None


def depth_normal_loss(depth: Tensor, gt_depth: Tensor, normal: Tensor, gt_normal: Tensor, depth_mask=None):
    """The reference's `depth_loss` and `normal_loss` (nerfstudio/models/gaussian_splatting.py:879-880) in one pass
    each way, without gathering the masked pixels:
        depth_loss  = F.l1_loss(depth[m], gt_depth[m])
        normal_loss = 0.5 F.mse_loss(normal[:, m], gt_normal[:, m]) + 0.5 cosine_similarity_loss(normal[:, m], gt_normal[:, m])
    depth (H, W, 1) and normal (H, W, 3) as the model returns them (channel slices of a wider image are read in
    place); gt_depth (H, W) / (1, H, W) / (H, W, 1); gt_normal (3, H, W) as the reference prepares it, or (H, W, 3);
    depth_mask (H, W) or (1, H, W) bool, None = every pixel.  Returns (depth_loss, normal_loss)."""
    if normal.dim() != 3:
        raise ValueError("normal must be (H, W, 3)")
    h, w = normal.shape[:2]
    if normal.shape[2] != 3 or depth.numel() != h * w or gt_depth.numel() != h * w or gt_normal.numel() != 3 * h * w:
        raise ValueError("expected depth (H, W, 1), gt_depth (H, W), normal (H, W, 3), gt_normal (3, H, W)")
    if depth_mask is not None and depth_mask.numel() != h * w:
        raise ValueError("depth_mask must be (H, W)")
    depth3 = depth if depth.dim() == 3 else depth.reshape(h, w, 1)
    gd = gt_depth.reshape(h, w) if gt_depth.is_contiguous() else gt_depth.contiguous().reshape(h, w)
    return _DepthNormalLoss.apply(depth3, gd, normal, gt_normal, depth_mask, h, w)


def gather_pixels(image: Tensor, *pixel_sets: Tensor):
    """Rows of a (H, W, C) image at several sets of (row, col) pixel coordinates — the reference's
    `feature[selected_pairs[i][0][:, 0], selected_pairs[i][0][:, 1]]` / `feature[selected_points[:, 0], ...]`
    (gaussian_splatting.py:912-917), one advanced-indexing gather per set there, each of whose backward allocates
    and zero-fills a whole (H, W, C) gradient image (61 MB at 1600x1200x32), scatters into it and is then added to
    the others.  Here ALL sets go through ONE gather (one index_select over the concatenated flat indices), so the
    backward is one zero-fill and one scatter.  Returns one (M_k, C) tensor per set.  Plain torch: no kernel."""
    h, w, c = image.shape
    flat = [(p[:, 0].long() * w + p[:, 1].long()) for p in pixel_sets]
    rows = image.reshape(h * w, c).index_select(0, torch.cat(flat))
    return torch.split(rows, [f.numel() for f in flat], dim=0)
