"""Host-side mirror of the gsplat-0.1.0 operator surface the reference model calls
(nerfstudio/models/gaussian_splatting.py:46-50 imports, :699/:730/:735/:747/:759/:773 calls),
sitting directly on the C ABI of libgg_raster.so (include/gg_raster.h).

Same class names, positional argument order, return shapes and error behaviour as
gsplat.project_gaussians.ProjectGaussians, gsplat.sh.SphericalHarmonics,
gsplat.rasterize.RasterizeGaussians and gsplat.nd_rasterize.NDRasterizeGaussians.
PyTorch is plumbing here: it owns device memory and the stream; every kernel is ours.

No CPU path: tensors must live on a HIP device ("cuda" in PyTorch-ROCm), otherwise the operators
raise.  One design difference from the reference, invisible to the caller: the four rasterize
calls of one view share ONE binning (tile sort), cached on the identity+version of
(xys, depths, radii, num_tiles_hit) — the reference re-sorts in every call.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional, Tuple

import torch
from torch import Tensor
from torch.autograd import Function

from . import _lib
from .constants import BLOCK, CLIP_THRESH_DEFAULT, deg_from_sh, num_sh_bases  # noqa: F401


def _require_hip(*tensors: Tensor) -> torch.device:
    dev = None
    for t in tensors:
        if not isinstance(t, Tensor):
            raise TypeError(f"expected a torch.Tensor, got {type(t)}")
        if t.device.type != "cuda":
            raise RuntimeError(
                "gaussiangrasper_amd operators run only on a HIP device (PyTorch-ROCm 'cuda'); "
                f"got a tensor on '{t.device}'. There is no CPU fallback.")
        if dev is None:
            dev = t.device
        elif t.device != dev:
            raise RuntimeError(f"tensors on different devices: {dev} vs {t.device}")
    return dev


def _f32(t: Tensor) -> Tensor:
    return t.contiguous() if t.dtype == torch.float32 else t.float().contiguous()


def _i32(t: Tensor) -> Tensor:
    return t.contiguous() if t.dtype == torch.int32 else t.int().contiguous()


def _ptr(t: Optional[Tensor]):
    return C.c_void_p(0 if t is None else t.data_ptr())


def _stream(dev: torch.device):
    return C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)


def _workspace(nbytes: int, dev: torch.device) -> Tensor:
    return torch.empty(max(int(nbytes), 256), dtype=torch.uint8, device=dev)


# ------------------------------------------------------------------------------------------------
# direct gradient accumulation (opt-in; used by dist.GradBucket)
# ------------------------------------------------------------------------------------------------
# Over the views of an optimizer step autograd adds every view's parameter gradient into `.grad`: for
# the two large parameters — SH coefficients (300 B per Gaussian) and features (128 B) — that is a
# freshly written gradient tensor plus a separate read-modify-write add per view.  When the caller has
# registered a persistent gradient buffer for a leaf parameter that enters an operator directly
# (`colors_all` into SphericalHarmonics, `feature` into NDRasterizeGaussians, as at reference
# :730,:747-753), the backward kernel adds into that buffer itself and returns no gradient for the
# input.  Off unless a sink is registered: plain autograd semantics otherwise.
_grad_sinks = {}        # id(param) -> (param, buffer, notify, defer)
_deferred_sh = {}       # id(param) -> [(degrees_to_use, num_bases, viewdirs, v_rgb), ...] waiting for expansion


def register_grad_sink(param: Tensor, buffer: Tensor, notify=None, defer=None) -> None:
    """Let the backward kernels accumulate the gradient of leaf `param` straight into `buffer`
    (same shape, fp32, contiguous); `notify(param)` is called after each accumulation is enqueued.
    `defer` (SH coefficients through ShadeTail only): a callable telling whether MORE contributions to this buffer
    will follow before it is read (the caller is in the middle of the views of an optimizer step).  While it
    returns True the operator keeps a view's SH gradient as its factors (12 B per Gaussian) instead of adding
    300 B per Gaussian into the buffer, and expands all kept views in one pass when it returns False — or when
    `flush_grad_sinks()` is called.  `notify` then fires once, after the expansion."""
    if not (param.is_leaf and param.requires_grad):
        raise ValueError("a gradient sink needs a leaf tensor that requires grad")
    if buffer.shape != param.shape or buffer.dtype != torch.float32 or not buffer.is_contiguous():
        raise ValueError("sink buffer must be a contiguous fp32 tensor of the parameter's shape")
    _grad_sinks[id(param)] = (param, buffer, notify, defer)


def clear_grad_sinks() -> None:
    _grad_sinks.clear()
    _deferred_sh.clear()


def discard_deferred_grads() -> None:
    """Forget kept-but-not-expanded contributions (the caller zeroes the gradients they belong to)."""
    _deferred_sh.clear()


def flush_grad_sinks() -> None:
    """Expand every kept SH contribution into its buffer now (gg_sh_bwd_multi) and fire the sinks' `notify`."""
    for key in list(_deferred_sh):
        pending = _deferred_sh.pop(key)
        sink = _grad_sinks.get(key)
        if sink is None or not pending:
            continue
        param, buf, notify = sink[0], sink[1], sink[2]
        dev = buf.device
        lib = _lib.load()
        deg, k = pending[0][0], pending[0][1]
        n = buf.shape[0]
        nv = len(pending)
        vd = (C.c_void_p * nv)(*[_ptr(p[2]) for p in pending])
        vc = (C.c_void_p * nv)(*[_ptr(p[3]) for p in pending])
        _lib.check(lib.gg_sh_bwd_multi(n, k, deg, nv, vd, vc, _ptr(buf), 1, _stream(dev)), "gg_sh_bwd_multi")
        if notify is not None:
            notify(param)


def _sink_for(t: Tensor):
    hit = _grad_sinks.get(id(t))
    if hit is None or hit[0] is not t:
        return None
    return hit


# ------------------------------------------------------------------------------------------------
# intersection count: started right after the projection, read when the first rasterize call needs it
# ------------------------------------------------------------------------------------------------
# The reference reads the cumulative tile count with `.item()` inside every rasterize call (SURVEY
# a5): a full host<->device round trip with the GPU idle while the host then enqueues the sort.  Here
# the count kernel and an async copy to pinned memory are enqueued at the end of
# ProjectGaussians.forward, followed by an event; whatever the caller enqueues next (SH, activations)
# keeps the GPU busy, and bin_and_sort_gaussians only waits on that event.
_PIN_SLOTS = 64
_pin_ring = {}          # device index -> (pinned int64[_PIN_SLOTS], next slot)
_pending_counts = {}    # (data_ptr, version) of num_tiles_hit -> (event, pinned view, keep-alive)


def _start_count(num_tiles_hit: Tensor, total: Optional[Tensor] = None, depth_parts=None) -> None:
    """total: the count already on its way (gg_project_fwd_count / gg_view_fwd left it there); None: count with a launch of
    its own.  depth_parts: (min bits, max bits, number of pairs) gg_view_fwd left behind, for gg_bin_sort_dev_ex"""
    dev = num_tiles_hit.device
    lib = _lib.load()
    ring, nxt = _pin_ring.get(dev.index, (None, 0))
    if ring is None:
        ring = torch.empty(_PIN_SLOTS, dtype=torch.int64).pin_memory()
    slot = ring[nxt:nxt + 1]
    _pin_ring[dev.index] = (ring, (nxt + 1) % _PIN_SLOTS)
    if total is None:
        total = torch.empty(1, dtype=torch.int64, device=dev)
        _lib.check(lib.gg_count_intersects(num_tiles_hit.shape[0], _ptr(num_tiles_hit), _ptr(total), None, 0,
                                           _stream(dev)), "gg_count_intersects")
    slot.copy_(total, non_blocking=True)
    ev = torch.cuda.Event()
    ev.record(torch.cuda.current_stream(dev))
    if len(_pending_counts) > 8:      # views whose count was never consumed
        _pending_counts.clear()
    _pending_counts[(num_tiles_hit.data_ptr(), num_tiles_hit._version)] = (ev, slot, num_tiles_hit, total, depth_parts)


def _pending(num_tiles_hit: Tensor):
    hit = _pending_counts.get((num_tiles_hit.data_ptr(), num_tiles_hit._version))
    if hit is None or hit[2] is not num_tiles_hit:
        return None
    return hit


def _take_count(num_tiles_hit: Tensor):
    hit = _pending(num_tiles_hit)
    if hit is None:
        return None
    _pending_counts.pop((num_tiles_hit.data_ptr(), num_tiles_hit._version), None)
    ev, slot = hit[0], hit[1]
    ev.synchronize()
    return int(slot.item())


# ------------------------------------------------------------------------------------------------
# ProjectGaussians
# ------------------------------------------------------------------------------------------------
class ProjectGaussians(Function):
    """gsplat.project_gaussians.ProjectGaussians.  apply(means3d, scales, glob_scale, quats,
    viewmat, projmat, fx, fy, cx, cy, img_height, img_width, tile_bounds, clip_thresh=0.01)
    -> (xys, depths, radii, conics, num_tiles_hit, cov3d)   [reference call :699-713]"""

    @staticmethod
    def forward(ctx, means3d: Tensor, scales: Tensor, glob_scale: float, quats: Tensor,
                viewmat: Tensor, projmat: Tensor, fx: float, fy: float, cx: float, cy: float,
                img_height: int, img_width: int, tile_bounds: Tuple[int, int, int],
                clip_thresh: float = CLIP_THRESH_DEFAULT):
        # cotangents nobody produced (cov3d always, others by route) arrive as None, not as zero-filled tensors:
        # autograd's default materialisation was a 24 MB fill kernel per view for cov3d alone
        ctx.set_materialize_grads(False)
        if means3d.ndim != 2 or means3d.shape[1] != 3:
            raise ValueError("means3d must have dimensions (N, 3)")
        n = means3d.shape[0]
        if tuple(scales.shape) != (n, 3):
            raise ValueError("scales must have dimensions (N, 3)")
        if tuple(quats.shape) != (n, 4):
            raise ValueError("quats must have dimensions (N, 4)")
        if viewmat.numel() < 12 or projmat.numel() != 16:
            raise ValueError("viewmat must hold >= 12 and projmat exactly 16 elements")
        dev = _require_hip(means3d, scales, quats, viewmat, projmat)
        ctx.sink = _sink_for(means3d) if (means3d.dtype == torch.float32 and means3d.is_contiguous()) else None
        means3d, scales, quats = _f32(means3d), _f32(scales), _f32(quats)
        viewmat, projmat = _f32(viewmat), _f32(projmat)
        lib = _lib.load()
        cov3d = torch.empty(n, 6, dtype=torch.float32, device=dev)
        xys = torch.empty(n, 2, dtype=torch.float32, device=dev)
        depths = torch.empty(n, dtype=torch.float32, device=dev)
        radii = torch.empty(n, dtype=torch.int32, device=dev)
        conics = torch.empty(n, 3, dtype=torch.float32, device=dev)
        num_tiles_hit = torch.empty(n, dtype=torch.int32, device=dev)
        # the projection also leaves sum(num_tiles_hit) on the device (one launch instead of projection + count + fill)
        total = torch.empty(1, dtype=torch.int64, device=dev)
        cws = torch.empty(max(lib.gg_project_count_workspace(n) // 4, 1), dtype=torch.int32, device=dev)
        _lib.check(lib.gg_project_fwd_count(
            n, _ptr(means3d), _ptr(scales), float(glob_scale), _ptr(quats), _ptr(viewmat),
            _ptr(projmat), float(fx), float(fy), float(cx), float(cy), int(img_height),
            int(img_width), int(tile_bounds[0]), int(tile_bounds[1]), float(clip_thresh),
            _ptr(cov3d), _ptr(xys), _ptr(depths), _ptr(radii), _ptr(conics), _ptr(num_tiles_hit), _ptr(total),
            _ptr(cws), cws.numel() * 4, _stream(dev)), "gg_project_fwd_count")
        ctx.scalars = (float(glob_scale), float(fx), float(fy), float(cx), float(cy),
                       int(img_height), int(img_width))
        ctx.save_for_backward(means3d, scales, quats, viewmat, projmat, radii, conics)
        ctx.mark_non_differentiable(radii, num_tiles_hit)
        _start_count(num_tiles_hit, total)
        return xys, depths, radii, conics, num_tiles_hit, cov3d

    @staticmethod
    def backward(ctx, v_xys, v_depths, v_radii, v_conics, v_num_tiles_hit, v_cov3d):
        means3d, scales, quats, viewmat, projmat, radii, conics = ctx.saved_tensors
        glob_scale, fx, fy, cx, cy, img_height, img_width = ctx.scalars
        dev = means3d.device
        n = means3d.shape[0]
        # v_xys / v_conics usually are columns of the blend backward's interleaved record: read in place
        v_xys, xy_stride = (torch.zeros(n, 2, device=dev), 2) if v_xys is None else _rows_in_place(v_xys, 2)
        v_conics, conic_stride = (torch.zeros(n, 3, device=dev), 3) if v_conics is None \
            else _rows_in_place(v_conics, 3)
        v_depths = torch.zeros(n, device=dev) if v_depths is None else _f32(v_depths)
        v_scale = torch.empty(n, 3, dtype=torch.float32, device=dev)
        v_quat = torch.empty(n, 4, dtype=torch.float32, device=dev)
        sink = ctx.sink
        v_mean3d = sink[1] if sink is not None else torch.empty(n, 3, dtype=torch.float32, device=dev)
        lib = _lib.load()
        _lib.check(lib.gg_project_bwd_ex(
            n, _ptr(means3d), _ptr(scales), glob_scale, _ptr(quats), _ptr(viewmat), _ptr(projmat),
            fx, fy, cx, cy, img_height, img_width, _ptr(radii), _ptr(conics), _ptr(v_xys), xy_stride,
            _ptr(v_depths), _ptr(v_conics), conic_stride, _ptr(v_mean3d), 1 if sink is not None else 0,
            _ptr(v_scale), _ptr(v_quat), _stream(dev)), "gg_project_bwd_ex")
        if sink is not None:          # the means' gradient went straight into the registered buffer
            if sink[2] is not None:
                sink[2](sink[0])
            v_mean3d = None
        return (v_mean3d, v_scale, None, v_quat, None, None, None, None, None, None, None, None,
                None, None)


# ------------------------------------------------------------------------------------------------
# SphericalHarmonics
# ------------------------------------------------------------------------------------------------
class SphericalHarmonics(Function):
    """gsplat.sh.SphericalHarmonics.  apply(degrees_to_use, viewdirs (N,3), coeffs (N,K,3))
    -> colors (N,3)   [reference call :730]"""

    @staticmethod
    def forward(ctx, degrees_to_use: int, viewdirs: Tensor, coeffs: Tensor):
        n, k = coeffs.shape[0], coeffs.shape[-2]
        assert k >= num_sh_bases(degrees_to_use), "not enough SH bases for degrees_to_use"
        deg_from_sh(k)  # raises ValueError on an invalid basis count
        if coeffs.shape[-1] != 3 or tuple(viewdirs.shape) != (n, 3):
            raise ValueError("viewdirs must be (N, 3) and coeffs (N, K, 3)")
        dev = _require_hip(viewdirs, coeffs)
        viewdirs, coeffs = _f32(viewdirs), _f32(coeffs)
        colors = torch.empty(n, 3, dtype=torch.float32, device=dev)
        lib = _lib.load()
        _lib.check(lib.gg_sh_fwd(n, k, int(degrees_to_use), _ptr(viewdirs), _ptr(coeffs),
                                 _ptr(colors), _stream(dev)), "gg_sh_fwd")
        ctx.degrees_to_use, ctx.num_bases = int(degrees_to_use), k
        ctx.save_for_backward(viewdirs)
        ctx.sink = _sink_for(coeffs) if coeffs.dtype == torch.float32 else None
        return colors

    @staticmethod
    def backward(ctx, v_colors: Tensor):
        (viewdirs,) = ctx.saved_tensors
        dev, n = viewdirs.device, viewdirs.shape[0]
        v_colors = _f32(v_colors)
        lib = _lib.load()
        if ctx.sink is not None:      # add into the registered gradient buffer, hand autograd nothing
            param, buf, notify, defer = ctx.sink
            pending = _deferred_sh.get(id(param))
            if defer is not None and (defer() or pending):
                # keep (view directions, colour cotangent) and expand the step's views in one pass (see ShadeTail)
                if pending and (pending[0][0], pending[0][1]) != (ctx.degrees_to_use, ctx.num_bases):
                    flush_grad_sinks()
                _deferred_sh.setdefault(id(param), []).append((ctx.degrees_to_use, ctx.num_bases, viewdirs,
                                                               v_colors))
                if not defer():
                    flush_grad_sinks()
                return None, None, None
            _lib.check(lib.gg_sh_bwd_accumulate(n, ctx.num_bases, ctx.degrees_to_use, _ptr(viewdirs),
                                                _ptr(v_colors), _ptr(buf), _stream(dev)),
                       "gg_sh_bwd_accumulate")
            if notify is not None:
                notify(param)
            return None, None, None
        v_coeffs = torch.empty(n, ctx.num_bases, 3, dtype=torch.float32, device=dev)
        _lib.check(lib.gg_sh_bwd(n, ctx.num_bases, ctx.degrees_to_use, _ptr(viewdirs),
                                 _ptr(v_colors), _ptr(v_coeffs), _stream(dev)), "gg_sh_bwd")
        return None, None, v_coeffs


def _rows_in_place(t: Tensor, width: int):
    """(tensor, row stride in floats) for a (N, width) fp32 cotangent whose rows can be read where they are — a
    dense array or the columns of a wider record (the blend backward's interleaved gradient record); anything
    else is copied to a dense array first."""
    if t.dtype == torch.float32 and t.dim() == 2 and t.shape[1] == width and t.stride(1) == 1 \
            and t.stride(0) >= width and t.data_ptr() % 4 == 0:
        return t, t.stride(0)
    t = _f32(t).reshape(-1, width)
    return t, width


class ShadeTail(Function):
    """apply(degrees_to_use, viewdirs (N,3), coeffs (N,K,3), depths (N,), normals (N,3)) -> tail (N, 7) =
    [ clamp(SphericalHarmonics(...) + 0.5, 0, 1) | depth | normal ]: the 7-channel colour array of the plugin
    route's rasterize operator in one kernel each way (reference :730-731 clamp, :765 depth, :779 normal colours).
    The backward reads the cotangent rows in place (the blend backward's interleaved gradient record, any row
    stride) and adds the SH gradient into a registered gradient sink like SphericalHarmonics does."""

    @staticmethod
    def forward(ctx, degrees_to_use: int, viewdirs: Tensor, coeffs: Tensor, depths: Tensor, normals: Tensor):
        n, k = coeffs.shape[0], coeffs.shape[-2]
        assert k >= num_sh_bases(degrees_to_use), "not enough SH bases for degrees_to_use"
        deg_from_sh(k)
        if coeffs.shape[-1] != 3 or tuple(viewdirs.shape) != (n, 3):
            raise ValueError("viewdirs must be (N, 3) and coeffs (N, K, 3)")
        if tuple(depths.shape) != (n,) or tuple(normals.shape) != (n, 3):
            raise ValueError("depths must be (N,) and normals (N, 3)")
        dev = _require_hip(viewdirs, coeffs, depths, normals)
        viewdirs, coeffs, depths_c, normals_c = _f32(viewdirs), _f32(coeffs), _f32(depths), _f32(normals)
        tail = torch.empty(n, 7, dtype=torch.float32, device=dev)
        mask = torch.empty(n, dtype=torch.uint8, device=dev)
        _lib.check(_lib.load().gg_shade_tail_fwd(n, k, int(degrees_to_use), _ptr(viewdirs), _ptr(coeffs),
                                                 _ptr(depths_c), _ptr(normals_c), _ptr(tail), _ptr(mask),
                                                 _stream(dev)), "gg_shade_tail_fwd")
        ctx.degrees_to_use, ctx.num_bases = int(degrees_to_use), k
        ctx.save_for_backward(viewdirs, mask)
        ctx.sink = _sink_for(coeffs) if coeffs.dtype == torch.float32 else None
        return tail

    @staticmethod
    def backward(ctx, v_tail: Tensor):
        viewdirs, mask = ctx.saved_tensors
        dev, n = viewdirs.device, viewdirs.shape[0]
        if v_tail.dtype != torch.float32 or v_tail.stride(1) != 1 or v_tail.stride(0) < 7:
            v_tail = _f32(v_tail)      # rows of a wider record are read in place; anything else is copied
        stride = v_tail.stride(0)
        v_depths = torch.empty(n, dtype=torch.float32, device=dev)
        v_normals = torch.empty(n, 3, dtype=torch.float32, device=dev)
        lib = _lib.load()
        if ctx.sink is not None:
            param, buf, notify, defer = ctx.sink
            pending = _deferred_sh.get(id(param))
            if defer is not None and (defer() or pending):
                # keep this view's SH gradient as its factors; expand all kept views once (the last view of the
                # step, or flush_grad_sinks()).  Same degree / basis count as what is already kept, or flush first
                if pending and (pending[0][0], pending[0][1]) != (ctx.degrees_to_use, ctx.num_bases):
                    flush_grad_sinks()
                v_rgb = torch.empty(n, 3, dtype=torch.float32, device=dev)
                _lib.check(lib.gg_shade_tail_bwd_split(n, _ptr(v_tail), stride, _ptr(mask), _ptr(v_rgb),
                                                       _ptr(v_depths), _ptr(v_normals), _stream(dev)),
                           "gg_shade_tail_bwd_split")
                _deferred_sh.setdefault(id(param), []).append((ctx.degrees_to_use, ctx.num_bases, viewdirs, v_rgb))
                if not defer():
                    flush_grad_sinks()
                return None, None, None, v_depths, v_normals
            _lib.check(lib.gg_shade_tail_bwd(n, ctx.num_bases, ctx.degrees_to_use, _ptr(viewdirs), _ptr(v_tail),
                                             stride, _ptr(mask), _ptr(buf), 1, _ptr(v_depths), _ptr(v_normals),
                                             _stream(dev)), "gg_shade_tail_bwd")
            if notify is not None:
                notify(param)
            return None, None, None, v_depths, v_normals
        v_coeffs = torch.empty(n, ctx.num_bases, 3, dtype=torch.float32, device=dev)
        _lib.check(lib.gg_shade_tail_bwd(n, ctx.num_bases, ctx.degrees_to_use, _ptr(viewdirs), _ptr(v_tail), stride,
                                         _ptr(mask), _ptr(v_coeffs), 0, _ptr(v_depths), _ptr(v_normals),
                                         _stream(dev)), "gg_shade_tail_bwd")
        return None, None, v_coeffs, v_depths, v_normals


# ------------------------------------------------------------------------------------------------
# binning shared by the rasterize calls of one view
# ------------------------------------------------------------------------------------------------
class Binning:
    """Result of compute_cumulative_intersects + bin_and_sort_gaussians for one view.

    `num_intersects` may still be unknown to the host when the lists are already being built on the
    device (speculative capacity, below): `resolve()` waits for the asynchronous read-back of the
    count — by then the sort and the first blend kernel are queued behind it, so the GPU does not idle —
    and re-bins with an exact size in the rare case the capacity was too small."""
    __slots__ = ("num_intersects", "gaussian_ids_sorted", "tile_bins", "key", "keep", "_redo")

    def __init__(self, num_intersects, gaussian_ids_sorted, tile_bins, key, keep, redo=None):
        self.num_intersects = num_intersects
        self.gaussian_ids_sorted = gaussian_ids_sorted
        self.tile_bins = tile_bins
        self.key = key
        self.keep = keep  # the keyed tensors stay alive so data_ptr cannot be recycled
        self._redo = redo

    def resolve(self) -> bool:
        """Make `num_intersects` known.  Returns True if the lists had to be rebuilt (whatever was
        rendered from them must be rendered again)."""
        if self.num_intersects is not None:
            return False
        return self._redo(self)


_bin_cache: Optional[Binning] = None
bin_cache_stats = {"hits": 0, "misses": 0, "speculative": 0, "rebinned": 0}
_capacity_hint = {}     # device index -> list capacity that covered the views seen so far


def _bin_key(xys, depths, radii, num_tiles_hit, img_height, img_width):
    return tuple((t.data_ptr(), t._version, tuple(t.shape)) for t in
                 (xys, depths, radii, num_tiles_hit)) + (int(img_height), int(img_width))


def last_num_intersects(radii: Optional[Tensor] = None) -> Optional[int]:
    """Length of the tile lists of the view binned last, once known to the host (the rasterize operators resolve
    it before they return); None before any binning.  0 means nothing was visible — what the reference's
    `(self.radii).sum() == 0` (gaussian_splatting.py:714) asks with a host round trip BEFORE the render.
    With `radii` (the projection output of THIS call): None as well unless the lists binned last are this call's —
    a view whose operator did not bin must not be judged by the previous view's count."""
    if _bin_cache is None:
        return None
    if radii is not None and not any(t is radii for t in _bin_cache.keep):
        return None
    return _bin_cache.num_intersects


def clear_bin_cache() -> None:
    global _bin_cache
    _bin_cache = None


def _note_count(dev: torch.device, num_intersects: int) -> None:
    want = ((int(num_intersects * 1.25) >> 20) + 1) << 20        # 25 % headroom, whole Mi entries
    if want > _capacity_hint.get(dev.index, 0):
        _capacity_hint[dev.index] = want


def bin_and_sort_gaussians(xys: Tensor, depths: Tensor, radii: Tensor, num_tiles_hit: Tensor,
                           img_height: int, img_width: int, use_cache: bool = True,
                           speculative: bool = False) -> Binning:
    """Tile lists of one view: Gaussian ids tile-major / near-to-far, and per-tile [start,end).

    The reference reads the intersection count with `.item()` before it can size and launch the sort
    (SURVEY a5): the GPU drains while the host waits, then idles while the host enqueues.  Here, once a
    view has been seen, the lists are built with a capacity that covered the previous views (+25 %) and
    the kernels take the actual count from device memory (gg_bin_sort_dev); the count is read back
    asynchronously and only checked later (`Binning.resolve`).  `speculative=False` (the default for
    direct callers) returns with `num_intersects` known."""
    global _bin_cache
    dev = _require_hip(xys, depths, radii, num_tiles_hit)
    key = _bin_key(xys, depths, radii, num_tiles_hit, img_height, img_width)
    if use_cache and _bin_cache is not None and _bin_cache.key == key:
        bin_cache_stats["hits"] += 1
        if not speculative:
            _bin_cache.resolve()
        return _bin_cache
    bin_cache_stats["misses"] += 1
    lib = _lib.load()
    n = xys.shape[0]
    tiles_x = (img_width + BLOCK - 1) // BLOCK
    tiles_y = (img_height + BLOCK - 1) // BLOCK
    xys_c, depths_c = _f32(xys.detach()), _f32(depths.detach())
    radii_c, nth_c = _i32(radii), _i32(num_tiles_hit)
    tile_bins = torch.empty(tiles_x * tiles_y, 2, dtype=torch.int32, device=dev)
    keep = (xys, depths, radii, num_tiles_hit)

    def exact(num_intersects: int) -> Tensor:
        ids = torch.empty(max(num_intersects, 1), dtype=torch.int32, device=dev)
        ws = _workspace(lib.gg_bin_sort_workspace(n, num_intersects), dev)
        _lib.check(lib.gg_bin_sort(n, num_intersects, _ptr(xys_c), _ptr(depths_c), _ptr(radii_c),
                                   _ptr(nth_c), tiles_x, tiles_y, _ptr(ids), _ptr(tile_bins),
                                   None, _ptr(ws), ws.numel(), _stream(dev)), "gg_bin_sort")
        return ids[:num_intersects]

    pending = _pending(num_tiles_hit) if speculative else None
    cap = _capacity_hint.get(dev.index, 0)
    if pending is not None and cap > 0:
        # speculative: sized by what previous views needed, count taken on the device
        bin_cache_stats["speculative"] += 1
        total_dev = pending[3]
        ids = torch.empty(cap, dtype=torch.int32, device=dev)
        ws = _workspace(lib.gg_bin_sort_workspace(n, cap), dev)
        parts = pending[4]                # (min bits, max bits, pairs) from the projection, or None
        _lib.check(lib.gg_bin_sort_dev_ex(n, cap, _ptr(total_dev), _ptr(xys_c), _ptr(depths_c), _ptr(radii_c),
                                          _ptr(nth_c), tiles_x, tiles_y, _ptr(ids), _ptr(tile_bins),
                                          None, _ptr(ws), ws.numel(), _ptr(parts[0]) if parts else None,
                                          _ptr(parts[1]) if parts else None, parts[2] if parts else 0, _stream(dev)),
                   "gg_bin_sort_dev_ex")

        def redo(b: Binning) -> bool:
            count = _take_count(num_tiles_hit)
            if count is None:                              # read-back already consumed by another binning
                count = int(total_dev.item())
            _note_count(dev, count)
            b.num_intersects = count
            if count <= cap:
                b.gaussian_ids_sorted = ids[:count]
                return False
            bin_cache_stats["rebinned"] += 1
            b.gaussian_ids_sorted = exact(count)
            return True

        out = Binning(None, ids, tile_bins, key, keep, redo)
    else:
        num_intersects = _take_count(num_tiles_hit)       # started by ProjectGaussians.forward
        if num_intersects is None:                        # tensors that did not come from our projection
            total = torch.empty(1, dtype=torch.int64, device=dev)
            _lib.check(lib.gg_count_intersects(n, _ptr(nth_c), _ptr(total), None, 0, _stream(dev)),
                       "gg_count_intersects")
            num_intersects = int(total.item())
        _note_count(dev, num_intersects)
        out = Binning(num_intersects, exact(num_intersects), tile_bins, key, keep)
    if use_cache:
        _bin_cache = out
    return out


# ------------------------------------------------------------------------------------------------
# Rasterize
# ------------------------------------------------------------------------------------------------
def _rasterize_forward(ctx, xys, depths, radii, conics, num_tiles_hit, colors, opacity,
                       img_height, img_width, background, three_channel_only):
    if colors.dtype == torch.uint8:
        colors = colors.float() / 255
    if background is not None:
        assert background.shape[0] == colors.shape[-1], \
            f"incorrect shape of background color tensor, expected shape {colors.shape[-1]}"
    else:
        background = torch.ones(colors.shape[-1], dtype=torch.float32, device=colors.device)
    if xys.ndimension() != 2 or xys.size(1) != 2:
        raise ValueError("xys must have dimensions (N, 2)")
    if three_channel_only:
        if colors.ndimension() != 2 or colors.size(1) != 3:
            raise ValueError("colors must have dimensions (N, 3)")
    elif colors.ndimension() != 2:
        raise ValueError("colors must have dimensions (N, D)")
    if opacity.ndimension() != 2 or opacity.size(1) != 1:
        raise ValueError("opacity must have dimensions (N, 1)")
    dev = _require_hip(xys, depths, radii, conics, num_tiles_hit, colors, opacity, background)
    n, ch = xys.size(0), colors.size(1)
    # wide colour gradients can go straight into a registered buffer (32-channel rows stay dense)
    ctx.sink = _sink_for(colors) if (ch > 3 and colors.dtype == torch.float32 and colors.is_contiguous()) else None
    img_height, img_width = int(img_height), int(img_width)
    xys_c, conics_c = _f32(xys), _f32(conics)
    colors_c, opacity_c, background = _f32(colors), _f32(opacity), _f32(background)

    bins = bin_and_sort_gaussians(xys, depths, radii, num_tiles_hit, img_height, img_width, speculative=True)
    ctx.img = (img_height, img_width)
    ctx.opacity_shape = tuple(opacity.shape)
    lib = _lib.load()
    ws = _workspace(lib.gg_blend_workspace(n), dev)
    out_img = final_Ts = final_idx = None
    for attempt in range(2):
        if bins.num_intersects is not None and bins.num_intersects < 1:
            ctx.num_intersects = bins.num_intersects
            out_img = torch.ones(img_height, img_width, ch, device=dev) * background
            ctx.save_for_backward(xys_c, conics_c, colors_c, opacity_c)
            return out_img
        if out_img is None:
            out_img = torch.empty(img_height, img_width, ch, dtype=torch.float32, device=dev)
            final_Ts = torch.empty(img_height, img_width, dtype=torch.float32, device=dev)
            final_idx = torch.empty(img_height, img_width, dtype=torch.int32, device=dev)
        _lib.check(lib.gg_blend_fwd(ch, n, img_height, img_width, _ptr(bins.gaussian_ids_sorted),
                                    _ptr(bins.tile_bins), _ptr(xys_c), _ptr(conics_c), _ptr(colors_c),
                                    _ptr(opacity_c), _ptr(background), _ptr(out_img), _ptr(final_Ts),
                                    _ptr(final_idx), _ptr(ws), ws.numel(), _stream(dev)),
                   "gg_blend_fwd")
        # the count of a speculative binning is checked only now, with the blend already queued;
        # rebuilt lists (capacity too small: rare) mean blending once more
        if not bins.resolve():
            break
    ctx.num_intersects = bins.num_intersects
    if bins.num_intersects < 1:
        out_img = torch.ones(img_height, img_width, ch, device=dev) * background
        ctx.save_for_backward(xys_c, conics_c, colors_c, opacity_c)
        return out_img
    # ws holds the packed per-Gaussian records of this call: the backward reuses them
    ctx.save_for_backward(xys_c, conics_c, colors_c, opacity_c, background,
                          bins.gaussian_ids_sorted, bins.tile_bins, final_Ts, final_idx, ws)
    return out_img


_DETERMINISTIC = False
# True: the fused operator's forward walk (feature 32 | rgb + depth + normal) runs the exact-summation-order kernel —
# images bit-identical to the oracle's sequential fma chain (gg_blend_fwd_pair); False (default): the batched kernel with
# fp16 two-piece products (gg_blend_fwd_pair_fast): same final_T / final_idx / decisions, images to fp32 rounding
# (~1e-7 of the colours' range; BASELINE asks for 1e-5).  Like mlp.EXACT_ORDER.
EXACT_FORWARD = False


def set_exact_forward(on: bool = True) -> bool:
    global EXACT_FORWARD
    prev, EXACT_FORWARD = EXACT_FORWARD, bool(on)
    return prev


def set_fast_forward(on: bool = True) -> bool:
    return not set_exact_forward(not on)


def set_deterministic_backward(on: bool = True) -> bool:
    """Bit-reproducible blend gradients.  The default backward adds pixel contributions with float atomics,
    whose order changes from run to run (as in gsplat 0.1.0: csrc/backward.cu atomicAdd).  With this switch the
    kernels store the total of every (tile-list entry, quadrant) instead and a second pass sums each
    Gaussian's entries in list order (`gg_blend_bwd_deterministic`): same values to ~1 ulp of the largest term,
    identical bits on every run, at the price of 16 (C + 6) bytes of scratch per list entry and roughly twice
    the time.  Returns the previous setting."""
    global _DETERMINISTIC
    prev, _DETERMINISTIC = _DETERMINISTIC, bool(on)
    return prev


def _blend_bwd(lib, ch, n, img_height, img_width, ids_sorted, tile_bins, xys, conics, colors, opacity, background,
               final_Ts, final_idx, v_out, v_xy, v_conic, v_colors, v_opacity, gstride, cstride, ws, flags,
               num_intersects):
    dev = xys.device
    if not _DETERMINISTIC:
        _lib.check(lib.gg_blend_bwd(ch, n, img_height, img_width, _ptr(ids_sorted), _ptr(tile_bins), _ptr(xys),
                                    _ptr(conics), _ptr(colors), _ptr(opacity), _ptr(background), _ptr(final_Ts),
                                    _ptr(final_idx), _ptr(v_out), _ptr(v_xy), _ptr(v_conic), _ptr(v_colors),
                                    _ptr(v_opacity), gstride, cstride, _ptr(ws), ws.numel(), flags,
                                    _stream(dev)), "gg_blend_bwd")
        return
    det_ws = _workspace(lib.gg_blend_bwd_deterministic_workspace(n, ch, num_intersects), dev)
    _lib.check(lib.gg_blend_bwd_deterministic(ch, n, img_height, img_width, _ptr(ids_sorted), _ptr(tile_bins),
                                              _ptr(xys), _ptr(conics), _ptr(colors), _ptr(opacity),
                                              _ptr(background), _ptr(final_Ts), _ptr(final_idx), _ptr(v_out),
                                              _ptr(v_xy), _ptr(v_conic), _ptr(v_colors), _ptr(v_opacity), gstride,
                                              cstride, _ptr(ws), ws.numel(), flags, num_intersects, _ptr(det_ws),
                                              det_ws.numel(), _stream(dev)), "gg_blend_bwd_deterministic")


def _rasterize_backward(ctx, v_out_img):
    img_height, img_width = ctx.img
    if ctx.num_intersects < 1:
        xys, conics, colors, opacity = ctx.saved_tensors
        v_xy, v_conic = torch.zeros_like(xys), torch.zeros_like(conics)
        v_colors, v_opacity = torch.zeros_like(colors), torch.zeros_like(opacity)
    else:
        (xys, conics, colors, opacity, background, ids_sorted, tile_bins, final_Ts,
         final_idx, ws) = ctx.saved_tensors
        dev, n, ch = xys.device, xys.shape[0], colors.shape[1]
        v_out_img = _f32(v_out_img)
        # Gradient layout: the kernels add with float atomics, and what they cost is cache lines
        # touched per instruction — so the geometry gradients of a Gaussian sit in ONE interleaved
        # record {xy, conic, opacity} (+ the colours for <= 3 channels); the tensors handed back to
        # autograd are strided views of it.  32-channel colour rows stay dense (128-byte rows).
        # (records padded to 64 / 32 bytes so that none straddles a line measure the same.)
        flags = 1                      # ws holds the forward's packed records
        sink = getattr(ctx, "sink", None)
        if ch <= 3:
            rec_g = torch.empty(n, 6 + ch, dtype=torch.float32, device=dev)
            v_colors = rec_g[:, 6:]
            gstride = cstride = 6 + ch
        else:
            rec_g = torch.empty(n, 6, dtype=torch.float32, device=dev)
            if sink is not None:       # atomics add into the caller's gradient buffer: no memset, no add
                v_colors = sink[1]
                flags |= 2
            else:
                v_colors = torch.empty(n, ch, dtype=torch.float32, device=dev)   # own allocation: aligned rows
            gstride, cstride = 6, 0
        v_xy, v_conic, v_opacity = rec_g[:, 0:2], rec_g[:, 2:5], rec_g[:, 5:6]
        lib = _lib.load()
        _blend_bwd(lib, ch, n, img_height, img_width, ids_sorted, tile_bins, xys, conics, colors, opacity,
                   background, final_Ts, final_idx, v_out_img, v_xy, v_conic, v_colors, v_opacity, gstride,
                   cstride, ws, flags, ctx.num_intersects)
        if flags & 2:
            v_colors = None
            if sink[2] is not None:
                sink[2](sink[0])
    return (v_xy, None, None, v_conic, None, v_colors, v_opacity.reshape(ctx.opacity_shape),
            None, None, None)


class RasterizeGaussians(Function):
    """gsplat.rasterize.RasterizeGaussians (3 colour channels).  apply(xys, depths, radii, conics,
    num_tiles_hit, colors (N,3), opacity (N,1), img_height, img_width, background=None)
    -> out_img (H,W,3)   [reference calls :735-746 rgb, :759-770 depth, :773-784 normal]"""

    @staticmethod
    def forward(ctx, xys, depths, radii, conics, num_tiles_hit, colors, opacity, img_height,
                img_width, background=None):
        return _rasterize_forward(ctx, xys, depths, radii, conics, num_tiles_hit, colors, opacity,
                                  img_height, img_width, background, True)

    @staticmethod
    def backward(ctx, v_out_img):
        return _rasterize_backward(ctx, v_out_img)


class NDRasterizeGaussians(Function):
    """gsplat.nd_rasterize.NDRasterizeGaussians (any channel count).  Same signature with
    colors (N,D) -> out_img (H,W,D)   [reference call :747-758, D = 32 feature channels]"""

    @staticmethod
    def forward(ctx, xys, depths, radii, conics, num_tiles_hit, colors, opacity, img_height,
                img_width, background=None):
        return _rasterize_forward(ctx, xys, depths, radii, conics, num_tiles_hit, colors, opacity,
                                  img_height, img_width, background, False)

    @staticmethod
    def backward(ctx, v_out_img):
        return _rasterize_backward(ctx, v_out_img)


# ------------------------------------------------------------------------------------------------
# caller-side activations of a view in one kernel each way
# ------------------------------------------------------------------------------------------------
class ActivateGaussians(Function):
    """apply(means, log_scales, quats, opacities (N,1) logits, cam_pos (3,)) ->
    (scales (N,3), quats_n (N,4), opac (N,1), viewdirs (N,3), normals (N,3)): the reference's
    `torch.exp(scales)` (:701), `quats / quats.norm(dim=-1, keepdim=True)` (:703),
    `torch.sigmoid(opacities)` (:742), view directions (:727-728, no gradient) and `get_normals()`
    (:605-619) — ~12 torch launches forward and ~25 backward per view in the reference, one launch each
    way here.  Used by the plugin's fused model; the shim route leaves the caller's torch ops alone."""

    @staticmethod
    def forward(ctx, means, log_scales, quats, opacities, cam_pos):
        ctx.set_materialize_grads(False)   # undefined cotangents arrive as None (handled in backward)
        dev = _require_hip(means, log_scales, quats, opacities, cam_pos)
        n = means.shape[0]
        if tuple(log_scales.shape) != (n, 3) or tuple(quats.shape) != (n, 4) or opacities.numel() != n:
            raise ValueError("expected means (N,3), scales (N,3), quats (N,4), opacities (N,1)")
        m, s, q, o, c = _f32(means), _f32(log_scales), _f32(quats), _f32(opacities), _f32(cam_pos).reshape(-1)
        f = lambda *shape: torch.empty(*shape, dtype=torch.float32, device=dev)
        scales, quats_n, opac, viewdirs, normals = f(n, 3), f(n, 4), f(n, 1), f(n, 3), f(n, 3)
        axis = torch.empty(n, dtype=torch.int32, device=dev)
        _lib.check(_lib.load().gg_activate_fwd(n, _ptr(m), _ptr(s), _ptr(q), _ptr(o), _ptr(c), _ptr(scales),
                                               _ptr(quats_n), _ptr(opac), _ptr(viewdirs), _ptr(normals), _ptr(axis),
                                               _stream(dev)), "gg_activate_fwd")
        ctx.save_for_backward(q, scales, opac, axis)
        ctx.opacity_shape = tuple(opacities.shape)
        # all three registered, or none: one accumulate switch for the kernel
        sinks = [_sink_for(t) if (t.dtype == torch.float32 and t.is_contiguous()) else None
                 for t in (log_scales, quats, opacities)]
        ctx.sinks = sinks if all(k is not None for k in sinks) else None
        ctx.mark_non_differentiable(viewdirs)
        return scales, quats_n, opac.reshape(ctx.opacity_shape), viewdirs, normals

    @staticmethod
    def backward(ctx, v_scales, v_quats_n, v_opac, v_viewdirs, v_normals):
        q, scales, opac, axis = ctx.saved_tensors
        dev, n = q.device, q.shape[0]
        z = lambda t, *shape: torch.zeros(*shape, dtype=torch.float32, device=dev) if t is None else _f32(t)
        v_scales, v_quats_n = z(v_scales, n, 3), z(v_quats_n, n, 4)
        v_normals = z(v_normals, n, 3)
        # v_opac usually is column 5 of the blend backward's interleaved record: read in place
        v_opac, opac_stride = (torch.zeros(n, 1, device=dev), 1) if v_opac is None \
            else _rows_in_place(v_opac.reshape(n, 1), 1)
        if ctx.sinks is not None:     # add straight into the registered gradient buffers
            v_ls, v_q, v_o = (k[1] for k in ctx.sinks)
        else:
            v_ls = torch.empty(n, 3, dtype=torch.float32, device=dev)
            v_q = torch.empty(n, 4, dtype=torch.float32, device=dev)
            v_o = torch.empty(n, 1, dtype=torch.float32, device=dev)
        _lib.check(_lib.load().gg_activate_bwd_ex(
            n, _ptr(q), _ptr(scales), _ptr(opac), _ptr(axis), _ptr(v_scales), _ptr(v_quats_n), _ptr(v_opac),
            opac_stride, _ptr(v_normals), _ptr(v_ls), _ptr(v_q), _ptr(v_o), 1 if ctx.sinks is not None else 0,
            _stream(dev)), "gg_activate_bwd_ex")
        if ctx.sinks is not None:
            for param, _buf, notify, _defer in ctx.sinks:
                if notify is not None:
                    notify(param)
            return None, None, None, None, None
        return None, v_ls, v_q, v_o.reshape(ctx.opacity_shape), None


# ------------------------------------------------------------------------------------------------
# the per-Gaussian chain of a view as ONE autograd node (round 4; what the plugin's fused model calls)
# ------------------------------------------------------------------------------------------------
class _PartCtx:
    """The `ctx` of a Function whose forward / backward run INSIDE another Function (ViewGeometry): attributes are kept,
    saved tensors are handed to the outer ctx.save_for_backward (a tensor held as a plain attribute of a node that
    produced it would be a reference cycle)."""

    def __init__(self):
        self.saved_tensors = ()

    def save_for_backward(self, *tensors):
        self.saved_tensors = tensors

    def mark_non_differentiable(self, *tensors):
        pass

    def set_materialize_grads(self, value):
        pass


def _is_record_view(t: Optional[Tensor], base: int, offset: int, width: int, stride: int, n: int) -> bool:
    return (t is not None and t.dtype == torch.float32 and t.dim() == 2 and tuple(t.shape) == (n, width)
            and t.stride(1) == 1 and t.stride(0) == stride and t.data_ptr() == base + 4 * offset)


class ViewGeometry(Function):
    """apply(means, log_scales, quats, opacities, colors_all, cam_pos, viewmat, full_proj, fx, fy, cx, cy,
             img_height, img_width, tile_bounds, degrees_to_use)
       -> (xys, depths, radii, conics, num_tiles_hit, opac, tail (N, 7), normals, packed)

    `packed`: a blend workspace already holding these Gaussians' packed records (pass it to rasterize_segments(...,
    packed=...): the forward then skips its packing pass).

    ActivateGaussians -> ProjectGaussians -> ShadeTail (reference :699-731, :742, :605-619) as one node: the forward is
    those three operators' forwards, the backward — when every parameter has a gradient buffer (register_grad_sink) and
    the cotangents arrive as the columns of the blend backward's per-Gaussian record, i.e. in the plugin's training
    step — is ONE kernel (gg_view_bwd) instead of three that hand (N, k) arrays to each other; the SH gradient is kept
    or expanded exactly as ShadeTail does.  Anything else (no sinks, dense cotangents, extra cotangents for depths /
    normals) runs the three operators' backwards in order: same values either way (shared device code)."""

    @staticmethod
    def forward(ctx, means, log_scales, quats, opacities, colors_all, cam_pos, viewmat, full_proj, fx, fy, cx, cy,
                img_height, img_width, tile_bounds, degrees_to_use):
        ctx.set_materialize_grads(False)
        a, p, t = _PartCtx(), _PartCtx(), _PartCtx()
        scales_e, quats_n, opac, viewdirs, normals, xys, depths, radii, conics, num_tiles_hit, packed = \
            ViewGeometry._activate_and_project(a, p, means, log_scales, quats, opacities, cam_pos, viewmat, full_proj,
                                               fx, fy, cx, cy, img_height, img_width, tile_bounds)
        tail = ShadeTail.forward(t, degrees_to_use, viewdirs, colors_all, depths, normals)
        ctx.parts = (a, p, t)
        ctx.counts = tuple(len(c.saved_tensors) for c in ctx.parts)
        ctx.save_for_backward(*a.saved_tensors, *p.saved_tensors, *t.saved_tensors)
        for c in ctx.parts:
            c.saved_tensors = ()
        ctx.mark_non_differentiable(radii, num_tiles_hit, packed)
        return xys, depths, radii, conics, num_tiles_hit, opac, tail, normals, packed

    @staticmethod
    def _activate_and_project(a, p, means, log_scales, quats, opacities, cam_pos, viewmat, full_proj, fx, fy, cx, cy,
                              img_height, img_width, tile_bounds):
        """ActivateGaussians.forward + ProjectGaussians.forward (glob_scale 1, default clip) as ONE kernel, gg_view_fwd,
        which also leaves the intersection count and the depth range's partial results behind; `a` / `p` are filled as
        the two forwards fill their ctx (the backwards read them)."""
        dev = _require_hip(means, log_scales, quats, opacities, cam_pos, viewmat, full_proj)
        n = means.shape[0]
        if means.ndim != 2 or means.shape[1] != 3:
            raise ValueError("means3d must have dimensions (N, 3)")
        if tuple(log_scales.shape) != (n, 3) or tuple(quats.shape) != (n, 4) or opacities.numel() != n:
            raise ValueError("expected means (N,3), scales (N,3), quats (N,4), opacities (N,1)")
        if viewmat.numel() < 12 or full_proj.numel() != 16:
            raise ValueError("viewmat must hold >= 12 and projmat exactly 16 elements")
        a.sinks = [_sink_for(t_) if (t_.dtype == torch.float32 and t_.is_contiguous()) else None
                   for t_ in (log_scales, quats, opacities)]
        a.sinks = a.sinks if all(k is not None for k in a.sinks) else None
        p.sink = _sink_for(means) if (means.dtype == torch.float32 and means.is_contiguous()) else None
        m, s_, q, o = _f32(means), _f32(log_scales), _f32(quats), _f32(opacities)
        c, vm, pm = _f32(cam_pos).reshape(-1), _f32(viewmat), _f32(full_proj)
        f = lambda *shape: torch.empty(*shape, dtype=torch.float32, device=dev)
        i32 = lambda *shape: torch.empty(*shape, dtype=torch.int32, device=dev)
        scales_e, quats_n, opac, viewdirs, normals = f(n, 3), f(n, 4), f(n, 1), f(n, 3), f(n, 3)
        xys, depths, conics = f(n, 2), f(n), f(n, 3)
        axis, radii, num_tiles_hit = i32(n), i32(n), i32(n)
        lib = _lib.load()
        blocks = max((n + 255) // 256, 1)
        parts = i32(3 * blocks)
        total = torch.empty(1, dtype=torch.int64, device=dev)
        packed = _workspace(lib.gg_blend_workspace(n), dev)          # the blend operators' workspace, records packed here
        _lib.check(lib.gg_view_fwd(
            n, _ptr(m), _ptr(s_), _ptr(q), _ptr(o), _ptr(c), _ptr(vm), _ptr(pm), float(fx), float(fy), float(cx),
            float(cy), int(img_height), int(img_width), int(tile_bounds[0]), int(tile_bounds[1]), CLIP_THRESH_DEFAULT,
            _ptr(scales_e), _ptr(quats_n), _ptr(opac), _ptr(viewdirs), _ptr(normals), _ptr(axis), _ptr(xys),
            _ptr(depths), _ptr(radii), _ptr(conics), _ptr(num_tiles_hit), _ptr(total), _ptr(parts), parts.numel() * 4,
            _ptr(packed), packed.numel(), _stream(dev)), "gg_view_fwd")
        a.save_for_backward(q, scales_e, opac, axis)
        a.opacity_shape = tuple(opacities.shape)
        p.scalars = (1.0, float(fx), float(fy), float(cx), float(cy), int(img_height), int(img_width))
        p.save_for_backward(m, scales_e, quats_n, vm, pm, radii, conics)
        _start_count(num_tiles_hit, total, (parts[blocks:2 * blocks], parts[2 * blocks:], blocks) if n > 0 else None)
        return (scales_e, quats_n, opac.reshape(a.opacity_shape), viewdirs, normals, xys, depths, radii, conics,
                num_tiles_hit, packed)

    @staticmethod
    def backward(ctx, v_xys, v_depths, v_radii, v_conics, v_nth, v_opac, v_tail, v_normals, v_packed=None):
        a, p, t = ctx.parts
        saved, k = ctx.saved_tensors, 0
        for c, cnt in zip(ctx.parts, ctx.counts):
            c.saved_tensors = saved[k:k + cnt]
            k += cnt
        q_raw, scales_e, opac, axis = a.saved_tensors
        means, _scales, quats_n, viewmat, projmat, radii, conics = p.saved_tensors
        viewdirs, mask = t.saved_tensors
        dev, n = means.device, means.shape[0]
        sinks_ok = a.sinks is not None and p.sink is not None and t.sink is not None
        fast = sinks_ok and v_depths is None and v_normals is None and v_xys is not None
        if fast:
            base, stride = v_xys.data_ptr(), v_xys.stride(0)
            fast = (stride >= 13 and _is_record_view(v_xys, base, 0, 2, stride, n)
                    and _is_record_view(v_conics, base, 2, 3, stride, n)
                    and _is_record_view(v_opac.reshape(n, 1) if v_opac is not None and v_opac.numel() == n else None,
                                        base, 5, 1, stride, n)
                    and _is_record_view(v_tail, base, 6, 7, stride, n)
                    and (stride % 4 != 0 or base % 16 == 0))
        if not fast:
            g_t = ShadeTail.backward(t, v_tail if v_tail is not None else torch.zeros(n, 7, device=dev))
            vd = g_t[3] if v_depths is None else g_t[3] + _f32(v_depths)
            vn = g_t[4] if v_normals is None else g_t[4] + _f32(v_normals)
            g_p = ProjectGaussians.backward(p, v_xys, vd, None, v_conics, None, None)
            g_a = ActivateGaussians.backward(a, g_p[1], g_p[3], v_opac, None, vn)
            return (g_p[0], g_a[1], g_a[2], g_a[3], g_t[2], None, None, None, None, None, None, None, None, None,
                    None, None)
        lib = _lib.load()
        glob_scale, fx, fy, _cx, _cy, img_height, img_width = p.scalars
        v_rgb = torch.empty(n, 3, dtype=torch.float32, device=dev)
        v_ls, v_q, v_o = (s_[1] for s_ in a.sinks)
        _lib.check(lib.gg_view_bwd(
            n, C.c_void_p(base), stride, _ptr(mask), _ptr(means), _ptr(scales_e), glob_scale, _ptr(q_raw), _ptr(quats_n),
            _ptr(opac), _ptr(axis), _ptr(viewmat), _ptr(projmat), fx, fy, img_height, img_width, _ptr(radii),
            _ptr(conics), _ptr(v_rgb), _ptr(p.sink[1]), _ptr(v_ls), _ptr(v_q), _ptr(v_o), _stream(dev)), "gg_view_bwd")
        # the SH gradient: kept as its factors over the views of a step, or expanded now — ShadeTail.backward's rules
        param, buf, notify, defer = t.sink
        pending = _deferred_sh.get(id(param))
        if pending and (pending[0][0], pending[0][1]) != (t.degrees_to_use, t.num_bases):
            flush_grad_sinks()
        _deferred_sh.setdefault(id(param), []).append((t.degrees_to_use, t.num_bases, viewdirs, v_rgb))
        if defer is None or not defer():
            flush_grad_sinks()              # (expands what is kept — this view included — and notifies the sink)
        if p.sink[2] is not None:
            p.sink[2](p.sink[0])
        for param_, _buf, notify_, _defer in a.sinks:
            if notify_ is not None:
                notify_(param_)
        return (None,) * 16


# ------------------------------------------------------------------------------------------------
# several colour arrays from one binning (what the plugin's fused model calls)
# ------------------------------------------------------------------------------------------------
class RasterizeSegments(Function):
    """apply(xys, depths, radii, conics, num_tiles_hit, opacity, img_height, img_width,
             colors_0, background_0, colors_1, background_1, ...) -> (image_0, image_1, ...)

    The blend of K colour arrays (N, C_k) over the SAME Gaussians in one operator: one binning, one
    record packing, one `final_T / final_idx`, ONE set of geometry gradients (v_xy, v_conic, v_opacity
    accumulated by every segment's backward kernel into one record, GG_BWD_ACCUMULATE_GEOM) — what the
    reference's four rasterize calls (:735-784) compute, without their redundancy and without
    concatenating the colours: each array keeps its own storage (a 32-channel feature array keeps its
    128-byte aligned rows and, as a leaf, its gradient sink), each image is its own tensor.
    Images are bit-identical to NDRasterizeGaussians on each array."""

    @staticmethod
    def forward(ctx, xys, depths, radii, conics, num_tiles_hit, opacity, img_height, img_width, splits, *segs):
        if len(segs) < 2 or len(segs) % 2:
            raise ValueError("expected colors_0, background_0[, colors_1, background_1, ...]")
        if xys.ndimension() != 2 or xys.size(1) != 2:
            raise ValueError("xys must have dimensions (N, 2)")
        if opacity.ndimension() != 2 or opacity.size(1) != 1:
            raise ValueError("opacity must have dimensions (N, 1)")
        cols, bgs = list(segs[0::2]), list(segs[1::2])
        dev = _require_hip(xys, depths, radii, conics, num_tiles_hit, opacity, *cols, *bgs)
        n = xys.size(0)
        img_height, img_width = int(img_height), int(img_width)
        for c, b in zip(cols, bgs):
            if c.ndimension() != 2 or c.size(0) != n:
                raise ValueError("colors must have dimensions (N, D)")
            assert b.shape[0] == c.shape[1], f"incorrect shape of background color tensor, expected shape {c.shape[1]}"
        xys_c, conics_c, opacity_c = _f32(xys), _f32(conics), _f32(opacity)
        cols_c, bgs_c = [_f32(c) for c in cols], [_f32(b) for b in bgs]
        ctx.sinks = [(_sink_for(c) if (c.shape[1] > 8 and c.dtype == torch.float32 and c.is_contiguous()) else None)
                     for c in cols]
        bins = bin_and_sort_gaussians(xys, depths, radii, num_tiles_hit, img_height, img_width, speculative=True)
        lib = _lib.load()
        # splits may come as (splits, packed): `packed` = a blend workspace that already holds these Gaussians' records
        # (ViewGeometry's last output): the pair forward then skips its packing pass
        packed = None
        if isinstance(splits, tuple) and len(splits) == 2 and (splits[1] is None or isinstance(splits[1], Tensor)) \
                and isinstance(splits[0], tuple):
            splits, packed = splits
        if packed is not None and (packed.dtype != torch.uint8 or packed.numel() < lib.gg_blend_workspace(n)
                                   or packed.device != dev):
            raise ValueError("packed: a uint8 blend workspace of gg_blend_workspace(N) bytes on the operands' device")
        ws = packed if packed is not None else _workspace(lib.gg_blend_workspace(n), dev)
        final_Ts = torch.empty(img_height, img_width, dtype=torch.float32, device=dev)
        final_idx = torch.empty(img_height, img_width, dtype=torch.int32, device=dev)
        outs = [torch.empty(img_height, img_width, c.shape[1], dtype=torch.float32, device=dev) for c in cols_c]
        for attempt in range(2):
            if bins.num_intersects is not None and bins.num_intersects < 1:
                break
            # a >= 32-channel array carries a <= 8-channel one through its first forward walk
            wide = next((i for i, c in enumerate(cols_c) if c.shape[1] >= 32), None)
            small = next((i for i, c in enumerate(cols_c) if c.shape[1] <= 8), None) if wide is not None else None
            if small is not None and packed is not None:
                _lib.check(lib.gg_blend_fwd_pair_packed(
                    cols_c[wide].shape[1], cols_c[small].shape[1], n, img_height, img_width,
                    _ptr(bins.gaussian_ids_sorted), _ptr(bins.tile_bins), _ptr(cols_c[wide]), _ptr(cols_c[small]),
                    _ptr(bgs_c[wide]), _ptr(bgs_c[small]), _ptr(outs[wide]), _ptr(outs[small]), _ptr(final_Ts),
                    _ptr(final_idx), _ptr(ws), ws.numel(), 0 if (EXACT_FORWARD or _DETERMINISTIC) else 1,
                    _stream(dev)), "gg_blend_fwd_pair_packed")
            elif small is not None:
                pair_fwd = lib.gg_blend_fwd_pair if (EXACT_FORWARD or _DETERMINISTIC) else lib.gg_blend_fwd_pair_fast
                _lib.check(pair_fwd(
                    cols_c[wide].shape[1], cols_c[small].shape[1], n, img_height, img_width,
                    _ptr(bins.gaussian_ids_sorted), _ptr(bins.tile_bins), _ptr(xys_c), _ptr(conics_c),
                    _ptr(cols_c[wide]), _ptr(cols_c[small]), _ptr(opacity_c), _ptr(bgs_c[wide]), _ptr(bgs_c[small]),
                    _ptr(outs[wide]), _ptr(outs[small]), _ptr(final_Ts), _ptr(final_idx), _ptr(ws), ws.numel(),
                    _stream(dev)), "gg_blend_fwd_pair")
            for i, (c, b, o) in enumerate(zip(cols_c, bgs_c, outs)):
                if small is not None and i in (wide, small):
                    continue
                _lib.check(lib.gg_blend_fwd(c.shape[1], n, img_height, img_width, _ptr(bins.gaussian_ids_sorted),
                                            _ptr(bins.tile_bins), _ptr(xys_c), _ptr(conics_c), _ptr(c),
                                            _ptr(opacity_c), _ptr(b), _ptr(o), _ptr(final_Ts), _ptr(final_idx),
                                            _ptr(ws), ws.numel(), _stream(dev)), "gg_blend_fwd")
            if not bins.resolve():
                break
        ctx.num_intersects = bins.num_intersects
        ctx.img = (img_height, img_width)
        ctx.opacity_shape = tuple(opacity.shape)
        ctx.nseg = len(cols_c)
        # splits: {segment index: channel counts} — that segment's image is returned as several images (views of
        # one buffer) and its cotangents come back separately (and are handed to the kernel as they are)
        splits = dict(splits or {})
        out_map = []
        for i, c in enumerate(cols_c):
            sizes = list(splits.get(i, [c.shape[1]]))
            if sum(sizes) != c.shape[1] or min(sizes) < 1:
                raise ValueError("split sizes must be positive and add up to the segment's channels")
            start = 0
            for sz in sizes:
                out_map.append((i, start, start + sz))
                start += sz
        ctx.out_map = out_map
        if bins.num_intersects < 1:
            outs = [torch.ones(img_height, img_width, c.shape[1], device=dev) * b for c, b in zip(cols_c, bgs_c)]
            ctx.save_for_backward(xys_c, conics_c, opacity_c, *cols_c)
        else:
            ctx.save_for_backward(xys_c, conics_c, opacity_c, bins.gaussian_ids_sorted, bins.tile_bins, final_Ts,
                                  final_idx, ws, *cols_c, *bgs_c)
        return tuple(outs[i] if (e - b) == outs[i].shape[2] else outs[i][..., b:e] for i, b, e in out_map)

    @staticmethod
    def backward(ctx, *v_parts):
        k = ctx.nseg
        img_height, img_width = ctx.img
        if ctx.num_intersects < 1:
            xys, conics, opacity = ctx.saved_tensors[:3]
            cols = ctx.saved_tensors[3:]
            grads = []
            for c in cols:
                grads += [torch.zeros_like(c), None]
            return (torch.zeros_like(xys), None, None, torch.zeros_like(conics), None,
                    torch.zeros_like(opacity).reshape(ctx.opacity_shape), None, None, None, *grads)
        xys, conics, opacity, ids_sorted, tile_bins, final_Ts, final_idx, ws = ctx.saved_tensors[:8]
        cols, bgs = ctx.saved_tensors[8:8 + k], ctx.saved_tensors[8 + k:8 + 2 * k]
        dev, n = xys.device, xys.shape[0]
        lib = _lib.load()
        # one record per Gaussian for the geometry gradients of ALL segments; the colours of the
        # narrowest small segment (<= 8 channels) ride in the same record (one cache line per Gaussian)
        small = [i for i in range(k) if cols[i].shape[1] <= 8]
        rider = min(small, key=lambda i: cols[i].shape[1]) if small else None
        gwidth = 6 + (cols[rider].shape[1] if rider is not None else 0)
        wide = next((i for i in range(k) if cols[i].shape[1] >= 32), None)
        # The pair walk: 16-float records on a 64-byte boundary — geometry sums and the rider's colour gradients of a
        # Gaussian then leave the kernel as ONE atomic request (csrc/blend2.hip, MG); other walks: dense records
        gstride = 16 if (rider is not None and wide is not None and not _DETERMINISTIC and gwidth <= 16) else gwidth
        rec_g = torch.empty(n, gstride, dtype=torch.float32, device=dev)
        v_xy, v_conic, v_opacity = rec_g[:, 0:2], rec_g[:, 2:5], rec_g[:, 5:6]
        order = ([rider] if rider is not None else []) + [i for i in range(k) if i != rider]
        grads = [None] * k
        first = True

        def parts(i):
            """cotangents of segment i as [(image, channels)] in channel order"""
            out = []
            for (seg, b, e), v in zip(ctx.out_map, v_parts):
                if seg != i:
                    continue
                if v is None:
                    v = torch.zeros(img_height, img_width, e - b, dtype=torch.float32, device=dev)
                out.append((_f32(v), e - b))
            return out

        def cotangent(i):
            p = parts(i)
            return p[0][0] if len(p) == 1 else torch.cat([v for v, _ in p], dim=-1)

        # a >= 32-channel array carries the rider through its first backward walk (gg_blend_bwd_pair): one
        # walk computes alpha, T and the geometry gradients of both arrays
        if rider is not None and wide is not None and not _DETERMINISTIC:
            sink = ctx.sinks[wide]
            flags = 1
            if sink is not None:
                v_colors = sink[1]
                flags |= 2
            else:
                v_colors = torch.empty(n, cols[wide].shape[1], dtype=torch.float32, device=dev)
            vo_w, rp = cotangent(wide), parts(rider)
            if len(rp) > 3:
                rp = [(cotangent(rider), cols[rider].shape[1])]
            part_ptrs = (C.c_void_p * len(rp))(*[_ptr(v) for v, _ in rp])
            part_chs = (C.c_int * len(rp))(*[c for _, c in rp])
            _lib.check(lib.gg_blend_bwd_pair(
                cols[wide].shape[1], cols[rider].shape[1], n, img_height, img_width, _ptr(ids_sorted),
                _ptr(tile_bins), _ptr(xys), _ptr(conics), _ptr(cols[wide]), _ptr(cols[rider]), _ptr(opacity),
                _ptr(bgs[wide]), _ptr(bgs[rider]), _ptr(final_Ts), _ptr(final_idx), _ptr(vo_w), part_ptrs,
                part_chs, len(rp), _ptr(v_xy), _ptr(v_conic), _ptr(v_colors), _ptr(rec_g[:, 6:gwidth]),
                _ptr(v_opacity), gstride, 0, gstride, _ptr(ws), ws.numel(), flags, _stream(dev)),
                "gg_blend_bwd_pair")
            grads[rider] = rec_g[:, 6:gwidth]
            if flags & 2:
                if sink[2] is not None:
                    sink[2](sink[0])
            else:
                grads[wide] = v_colors
            order = [i for i in range(k) if i not in (wide, rider)]
            first = False
        for i in order:
            ch = cols[i].shape[1]
            v_out = cotangent(i)
            flags = 1 | (0 if first else 4)
            sink = ctx.sinks[i]
            if i == rider:
                v_colors, cstride = rec_g[:, 6:gwidth], gstride
            elif sink is not None:
                v_colors, cstride = sink[1], 0
                flags |= 2
            else:
                v_colors, cstride = torch.empty(n, ch, dtype=torch.float32, device=dev), 0
            _blend_bwd(lib, ch, n, img_height, img_width, ids_sorted, tile_bins, xys, conics, cols[i], opacity,
                       bgs[i], final_Ts, final_idx, v_out, v_xy, v_conic, v_colors, v_opacity, gstride, cstride,
                       ws, flags, ctx.num_intersects)
            first = False
            if flags & 2:
                if sink[2] is not None:
                    sink[2](sink[0])
            else:
                grads[i] = v_colors
        seg_grads = []
        for g in grads:
            seg_grads += [g, None]
        return (v_xy, None, None, v_conic, None, v_opacity.reshape(ctx.opacity_shape), None, None, None, *seg_grads)


def rasterize_segments(xys, depths, radii, conics, num_tiles_hit, opacity, img_height, img_width, segments,
                       packed: Optional[Tensor] = None):
    """segments: sequence of (colors (N, C_k), background (C_k,)[, split sizes]) -> list of images (H, W, C_k);
    a segment given with split sizes (e.g. [3, 1, 3] for rgb | depth | normal) comes back as that many images
    (views of one buffer, channel-sliced), and their cotangents go to the backward kernel without a concatenation."""
    flat, splits = [], {}
    for i, seg in enumerate(segments):
        flat += [seg[0], seg[1]]
        if len(seg) > 2 and seg[2] is not None:
            splits[i] = tuple(int(x) for x in seg[2])
    spec = tuple(sorted(splits.items()))
    return list(RasterizeSegments.apply(xys, depths, radii, conics, num_tiles_hit, opacity, img_height,
                                        img_width, (spec, packed) if packed is not None else spec, *flat))


# ------------------------------------------------------------------------------------------------
# quat_to_rotmat (gsplat._torch_impl; reference gaussian_splatting.py:516,614, scripts/update.py:204,229)
# ------------------------------------------------------------------------------------------------
class _QuatToRotmat(Function):
    @staticmethod
    def forward(ctx, quat: Tensor) -> Tensor:
        dev = _require_hip(quat)
        q = _f32(quat).reshape(-1, 4)
        rot = torch.empty(q.shape[0], 9, dtype=torch.float32, device=dev)
        _lib.check(_lib.load().gg_quat_to_rotmat_fwd(q.shape[0], _ptr(q), _ptr(rot), _stream(dev)),
                   "gg_quat_to_rotmat_fwd")
        ctx.save_for_backward(q)
        ctx.in_shape, ctx.in_dtype = quat.shape, quat.dtype
        return rot.reshape(quat.shape[:-1] + (3, 3))

    @staticmethod
    def backward(ctx, v_rot: Tensor):
        (q,) = ctx.saved_tensors
        dev = q.device
        g = _f32(v_rot).reshape(-1, 9)
        v_q = torch.empty_like(q)
        _lib.check(_lib.load().gg_quat_to_rotmat_bwd(q.shape[0], _ptr(q), _ptr(g), _ptr(v_q),
                                                     _stream(dev)), "gg_quat_to_rotmat_bwd")
        return v_q.reshape(ctx.in_shape).to(ctx.in_dtype)


def quat_to_rotmat(quat: Tensor) -> Tensor:
    """Rotation matrices (...,3,3) of wxyz quaternions (normalised first), differentiable.
    gsplat's is ~35 torch elementwise launches forward and ~70 backward on strided views; here
    one HIP kernel each way (gg_quat_to_rotmat_fwd/bwd)."""
    if quat.shape[-1] != 4:
        raise ValueError(f"quat must have dimensions (..., 4), got {tuple(quat.shape)}")
    return _QuatToRotmat.apply(quat)
