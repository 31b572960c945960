"""numpy front end of the CPU oracle (oracle/gg_oracle.c).  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module
(see the header of gg_oracle.c).  PARITY UNPINNED: the reference holds no fixture for this
path and its implementation (gsplat==0.1.0) is not available offline; the oracle restates
SURVEY.md §8a and is pinned only by the analytic / finite-difference tests in tests/.

Every function takes and returns numpy arrays in torch's layouts.  `dtype=np.float32` uses the
bit-exact parity build, `dtype=np.float64` the finite-difference build.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from typing import Tuple

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIBS = {}


def build(force: bool = False) -> None:
    """Compile both oracle libraries with the committed Makefile."""
    if force:
        subprocess.check_call(["make", "-C", _HERE, "clean"], stdout=subprocess.DEVNULL)
    subprocess.check_call(["make", "-C", _HERE], stdout=subprocess.DEVNULL)


VARIANT = ""        # "" = the default constants; "compat" = the recalled gsplat-0.1.0 deviations on


def use_variant(name: str) -> None:
    """Select which build of the oracle the functions below call ("" or "compat"; see the Makefile)."""
    global VARIANT
    assert name in ("", "compat")
    VARIANT = name


def _lib(dtype):
    key = ("compat_" if VARIANT == "compat" else "") + ("f64" if np.dtype(dtype) == np.float64 else "f32")
    if key not in _LIBS:
        path = os.path.join(_HERE, f"libgg_oracle_{key}.so")
        if not os.path.exists(path):
            build()
        _LIBS[key] = C.CDLL(path)
    f64 = key.endswith("f64")
    return _LIBS[key], ("ggo64_" if f64 else "ggo_"), (C.c_double if f64 else C.c_float)


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def _c(a, dtype):
    return np.ascontiguousarray(a, dtype=dtype)


def num_threads() -> int:
    lib, pre, _ = _lib(np.float32)
    return int(getattr(lib, pre + "num_threads")())


def set_num_threads(n: int) -> None:
    for dt in (np.float32, np.float64):
        lib, pre, _ = _lib(dt)
        getattr(lib, pre + "set_num_threads")(C.c_int(n))


def expf(x, dtype=np.float32):
    lib, pre, _ = _lib(dtype)
    x = _c(x, dtype)
    y = np.empty_like(x)
    getattr(lib, pre + "expf_array")(C.c_int(x.size), _p(x), _p(y))
    return y


def project_fwd(means, scales, glob_scale, quats, viewmat, projmat, fx, fy, cx, cy, img_h, img_w,
                tile_bounds, clip_thresh=0.01, dtype=np.float32):
    """-> (xys, depths, radii, conics, num_tiles_hit, cov3d), the order ProjectGaussians.apply
    returns them (nerfstudio/models/gaussian_splatting.py:699)."""
    lib, pre, RT = _lib(dtype)
    means, scales, quats = _c(means, dtype), _c(scales, dtype), _c(quats, dtype)
    viewmat, projmat = _c(viewmat, dtype).reshape(-1), _c(projmat, dtype).reshape(-1)
    assert viewmat.size >= 12 and projmat.size == 16
    n = means.shape[0]
    cov3d = np.zeros((n, 6), dtype)
    xys = np.zeros((n, 2), dtype)
    depths = np.zeros((n,), dtype)
    radii = np.zeros((n,), np.int32)
    conics = np.zeros((n, 3), dtype)
    nth = np.zeros((n,), np.int32)
    getattr(lib, pre + "project_fwd")(
        C.c_int(n), _p(means), _p(scales), RT(glob_scale), _p(quats), _p(viewmat), _p(projmat),
        RT(fx), RT(fy), RT(cx), RT(cy), C.c_int(img_h), C.c_int(img_w), C.c_int(tile_bounds[0]),
        C.c_int(tile_bounds[1]), RT(clip_thresh), _p(cov3d), _p(xys), _p(depths), _p(radii),
        _p(conics), _p(nth))
    return xys, depths, radii, conics, nth, cov3d


def project_bwd(means, scales, glob_scale, quats, viewmat, projmat, fx, fy, cx, cy, img_h, img_w,
                radii, conics, v_xy, v_depth, v_conic, dtype=np.float32):
    """-> (v_mean3d, v_scale, v_quat)"""
    lib, pre, RT = _lib(dtype)
    means, scales, quats = _c(means, dtype), _c(scales, dtype), _c(quats, dtype)
    viewmat, projmat = _c(viewmat, dtype).reshape(-1), _c(projmat, dtype).reshape(-1)
    radii = _c(radii, np.int32)
    conics, v_xy, v_depth, v_conic = (_c(conics, dtype), _c(v_xy, dtype), _c(v_depth, dtype),
                                      _c(v_conic, dtype))
    n = means.shape[0]
    vm, vs, vq = np.zeros((n, 3), dtype), np.zeros((n, 3), dtype), np.zeros((n, 4), dtype)
    getattr(lib, pre + "project_bwd")(
        C.c_int(n), _p(means), _p(scales), RT(glob_scale), _p(quats), _p(viewmat), _p(projmat),
        RT(fx), RT(fy), RT(cx), RT(cy), C.c_int(img_h), C.c_int(img_w), _p(radii), _p(conics),
        _p(v_xy), _p(v_depth), _p(v_conic), _p(vm), _p(vs), _p(vq))
    return vm, vs, vq


def sh_fwd(degrees_to_use, viewdirs, coeffs, dtype=np.float32):
    lib, pre, _ = _lib(dtype)
    viewdirs, coeffs = _c(viewdirs, dtype), _c(coeffs, dtype)
    n, k = coeffs.shape[0], coeffs.shape[1]
    out = np.zeros((n, 3), dtype)
    getattr(lib, pre + "sh_fwd")(C.c_int(n), C.c_int(k), C.c_int(degrees_to_use), _p(viewdirs),
                                 _p(coeffs), _p(out))
    return out


def sh_bwd(degrees_to_use, num_bases, viewdirs, v_colors, dtype=np.float32):
    lib, pre, _ = _lib(dtype)
    viewdirs, v_colors = _c(viewdirs, dtype), _c(v_colors, dtype)
    n = viewdirs.shape[0]
    out = np.zeros((n, num_bases, 3), dtype)
    getattr(lib, pre + "sh_bwd")(C.c_int(n), C.c_int(num_bases), C.c_int(degrees_to_use),
                                 _p(viewdirs), _p(v_colors), _p(out))
    return out


def shade_tail_fwd(degrees_to_use, viewdirs, coeffs, depths, normals, dtype=np.float32):
    """-> tail (N, 7) = [clamp(SH + 0.5, 0, 1) | depth | normal], clamp mask (N,) uint8"""
    lib, pre, _ = _lib(dtype)
    viewdirs, coeffs, depths, normals = _c(viewdirs, dtype), _c(coeffs, dtype), _c(depths, dtype), _c(normals, dtype)
    n, k = coeffs.shape[0], coeffs.shape[1]
    tail, mask = np.zeros((n, 7), dtype), np.zeros(n, np.uint8)
    getattr(lib, pre + "shade_tail_fwd")(C.c_int(n), C.c_int(k), C.c_int(degrees_to_use), _p(viewdirs), _p(coeffs),
                                         _p(depths), _p(normals), _p(tail), _p(mask))
    return tail, mask


def shade_tail_bwd(degrees_to_use, num_bases, viewdirs, v_tail, mask, v_coeffs_in=None, dtype=np.float32):
    """v_tail (N, stride >= 7) -> v_coeffs (N, K, 3) (added to v_coeffs_in when given), v_depths (N,), v_normals (N, 3)"""
    lib, pre, _ = _lib(dtype)
    viewdirs, v_tail, mask = _c(viewdirs, dtype), _c(v_tail, dtype), _c(mask, np.uint8)
    n = viewdirs.shape[0]
    out = np.zeros((n, num_bases, 3), dtype) if v_coeffs_in is None else _c(v_coeffs_in, dtype).copy()
    vd, vn = np.zeros(n, dtype), np.zeros((n, 3), dtype)
    getattr(lib, pre + "shade_tail_bwd")(C.c_int(n), C.c_int(num_bases), C.c_int(degrees_to_use), _p(viewdirs),
                                         _p(v_tail), C.c_int(v_tail.shape[1]), _p(mask), _p(out),
                                         C.c_int(0 if v_coeffs_in is None else 1), _p(vd), _p(vn))
    return out, vd, vn


def image_loss_fwd(rgb, gt, valid, ssim_lambda=0.2, dtype=np.float32):
    """-> (main_loss, Ll1, ssim) of get_loss_dict :882-885, :931; rgb, gt (H, W, 3); valid (H, W) bool or None"""
    lib, pre, RT = _lib(dtype)
    rgb, gt = _c(rgb, dtype), _c(gt, dtype)
    h, w = rgb.shape[:2]
    v = None if valid is None else _c(np.asarray(valid).astype(np.uint8), np.uint8)
    out = np.zeros(3, dtype)
    getattr(lib, pre + "image_loss_fwd")(C.c_int(h), C.c_int(w), _p(rgb), _p(gt), _p(v) if v is not None else None,
                                         RT(ssim_lambda), _p(out))
    return out


def image_loss_bwd(rgb, gt, valid, ssim_lambda=0.2, v_main=1.0, dtype=np.float32):
    lib, pre, RT = _lib(dtype)
    rgb, gt = _c(rgb, dtype), _c(gt, dtype)
    h, w = rgb.shape[:2]
    v = None if valid is None else _c(np.asarray(valid).astype(np.uint8), np.uint8)
    out = np.zeros((h, w, 3), dtype)
    getattr(lib, pre + "image_loss_bwd")(C.c_int(h), C.c_int(w), _p(rgb), _p(gt), _p(v) if v is not None else None,
                                         RT(ssim_lambda), RT(v_main), _p(out))
    return out


def _geom_args(depth, gt_depth, normal, gt_normal_chw, mask, dtype):
    depth, gt_depth = _c(depth, dtype).reshape(-1), _c(gt_depth, dtype).reshape(-1)
    normal, gtn = _c(normal, dtype), _c(gt_normal_chw, dtype)          # (H, W, 3) and (3, H, W)
    p = depth.shape[0]
    m = None if mask is None else _c(np.asarray(mask).astype(np.uint8).reshape(-1), np.uint8)
    args = [C.c_int64(p), _p(depth), C.c_int(1), _p(gt_depth), C.c_int(1), _p(normal), C.c_int(3), C.c_int(1),
            _p(gtn), C.c_int(1), C.c_int(p), _p(m) if m is not None else None]
    return p, args, (depth, gt_depth, normal, gtn, m)


def geom_loss_fwd(depth, gt_depth, normal, gt_normal_chw, mask, dtype=np.float32):
    """depth (H, W[, 1]), gt_depth (H, W), normal (H, W, 3), gt_normal (3, H, W), mask (H, W) -> (depth_loss,
    normal_loss, count) of get_loss_dict :879-880"""
    lib, pre, _ = _lib(dtype)
    p, args, keep = _geom_args(depth, gt_depth, normal, gt_normal_chw, mask, dtype)
    out = np.zeros(3, dtype)
    getattr(lib, pre + "geom_loss_fwd")(*args, _p(out))
    return out


def geom_loss_bwd(depth, gt_depth, normal, gt_normal_chw, mask, v_depth_loss=1.0, v_normal_loss=1.0, dtype=np.float32):
    lib, pre, RT = _lib(dtype)
    p, args, keep = _geom_args(depth, gt_depth, normal, gt_normal_chw, mask, dtype)
    vd, vn = np.zeros(p, dtype), np.zeros((p, 3), dtype)
    getattr(lib, pre + "geom_loss_bwd")(*args, RT(v_depth_loss), RT(v_normal_loss), _p(vd), _p(vn))
    return vd, vn


def bin_and_sort(xys, depths, radii, num_tiles_hit, tile_bounds, dtype=np.float32):
    """compute_cumulative_intersects + bin_and_sort_gaussians.
    -> dict(num_intersects, cum_tiles_hit, isect_ids, gaussian_ids, isect_ids_sorted,
            gaussian_ids_sorted, tile_bins)"""
    lib, pre, _ = _lib(dtype)
    xys, depths = _c(xys, dtype), _c(depths, dtype)
    radii, nth = _c(radii, np.int32), _c(num_tiles_hit, np.int32)
    n = xys.shape[0]
    cum = np.zeros((n,), np.int32)
    fn = getattr(lib, pre + "cumsum")
    fn.restype = C.c_int64
    total = int(fn(C.c_int(n), _p(nth), _p(cum)))
    tiles_x, tiles_y = int(tile_bounds[0]), int(tile_bounds[1])
    keys = np.zeros((total,), np.int64)
    ids = np.zeros((total,), np.int32)
    keys_s = np.zeros((total,), np.int64)
    ids_s = np.zeros((total,), np.int32)
    bins = np.zeros((tiles_x * tiles_y, 2), np.int32)
    if total > 0:
        getattr(lib, pre + "map_intersects")(C.c_int(n), _p(xys), _p(depths), _p(radii), _p(cum),
                                             C.c_int(tiles_x), C.c_int(tiles_y), _p(keys), _p(ids))
        getattr(lib, pre + "sort_intersects")(C.c_int64(total), _p(keys), _p(ids), _p(keys_s),
                                              _p(ids_s))
        getattr(lib, pre + "tile_bins")(C.c_int64(total), _p(keys_s), C.c_int(tiles_x * tiles_y),
                                        _p(bins))
    return dict(num_intersects=total, cum_tiles_hit=cum, isect_ids=keys, gaussian_ids=ids,
                isect_ids_sorted=keys_s, gaussian_ids_sorted=ids_s, tile_bins=bins)


def blend_fwd(ids_sorted, tile_bins, xys, conics, colors, opacity, img_h, img_w, background,
              dtype=np.float32) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
    """-> (out_img (H,W,C), final_Ts (H,W), final_idx (H,W))"""
    lib, pre, _ = _lib(dtype)
    ids_sorted, tile_bins = _c(ids_sorted, np.int32), _c(tile_bins, np.int32)
    xys, conics, colors = _c(xys, dtype), _c(conics, dtype), _c(colors, dtype)
    opacity, background = _c(opacity, dtype).reshape(-1), _c(background, dtype).reshape(-1)
    ch = colors.shape[1]
    assert background.size == ch and opacity.size == xys.shape[0]
    tiles_x, tiles_y = (img_w + 15) // 16, (img_h + 15) // 16
    assert tile_bins.shape[0] == tiles_x * tiles_y
    out = np.zeros((img_h, img_w, ch), dtype)
    ft = np.zeros((img_h, img_w), dtype)
    fi = np.zeros((img_h, img_w), np.int32)
    getattr(lib, pre + "blend_fwd")(
        C.c_int(ch), C.c_int(img_h), C.c_int(img_w), C.c_int(tiles_x), C.c_int(tiles_y),
        _p(ids_sorted), _p(tile_bins), _p(xys), _p(conics), _p(colors), _p(opacity),
        _p(background), _p(out), _p(ft), _p(fi))
    return out, ft, fi


def blend_bwd(ids_sorted, tile_bins, xys, conics, colors, opacity, img_h, img_w, background,
              final_Ts, final_idx, v_out, dtype=np.float32):
    """-> (v_xy, v_conic, v_colors, v_opacity (N,1))"""
    lib, pre, _ = _lib(dtype)
    ids_sorted, tile_bins = _c(ids_sorted, np.int32), _c(tile_bins, np.int32)
    xys, conics, colors = _c(xys, dtype), _c(conics, dtype), _c(colors, dtype)
    opacity, background = _c(opacity, dtype).reshape(-1), _c(background, dtype).reshape(-1)
    final_Ts, final_idx, v_out = _c(final_Ts, dtype), _c(final_idx, np.int32), _c(v_out, dtype)
    n, ch = colors.shape
    tiles_x, tiles_y = (img_w + 15) // 16, (img_h + 15) // 16
    v_xy, v_conic = np.zeros((n, 2), dtype), np.zeros((n, 3), dtype)
    v_colors, v_opacity = np.zeros((n, ch), dtype), np.zeros((n, 1), dtype)
    getattr(lib, pre + "blend_bwd")(
        C.c_int(ch), C.c_int(n), C.c_int(img_h), C.c_int(img_w), C.c_int(tiles_x),
        C.c_int(tiles_y), _p(ids_sorted), _p(tile_bins), _p(xys), _p(conics), _p(colors),
        _p(opacity), _p(background), _p(final_Ts), _p(final_idx), _p(v_out), _p(v_xy),
        _p(v_conic), _p(v_colors), _p(v_opacity))
    return v_xy, v_conic, v_colors, v_opacity


def rasterize_fwd(xys, depths, radii, conics, num_tiles_hit, colors, opacity, img_h, img_w,
                  background, dtype=np.float32):
    """Whole Rasterize*.forward: bin + sort + blend.  -> (out_img, saved dict)"""
    tb = ((img_w + 15) // 16, (img_h + 15) // 16, 1)
    b = bin_and_sort(xys, depths, radii, num_tiles_hit, tb, dtype)
    ch = np.asarray(colors).shape[1]
    if b["num_intersects"] < 1:
        bg = np.asarray(background, dtype).reshape(1, 1, ch)
        out = np.ones((img_h, img_w, ch), dtype) * bg
        return out, dict(bins=b, final_Ts=None, final_idx=None)
    out, ft, fi = blend_fwd(b["gaussian_ids_sorted"], b["tile_bins"], xys, conics, colors, opacity,
                            img_h, img_w, background, dtype)
    return out, dict(bins=b, final_Ts=ft, final_idx=fi)


def quat_to_rotmat(quats, dtype=np.float32):
    """gsplat._torch_impl.quat_to_rotmat (reference call sites gaussian_splatting.py:516,614;
    scripts/update.py:204,229): normalise wxyz, standard rotation matrix, (...,3,3)."""
    lib, pre, _ = _lib(dtype)
    q = np.ascontiguousarray(np.asarray(quats, dtype).reshape(-1, 4))
    rot = np.empty((q.shape[0], 9), dtype)
    getattr(lib, pre + "quat_to_rotmat_fwd")(C.c_int(q.shape[0]), _p(q), _p(rot))
    return rot.reshape(np.shape(quats)[:-1] + (3, 3))


def quat_to_rotmat_bwd(quats, v_rot, dtype=np.float32):
    """VJP of quat_to_rotmat: v_quats (...,4) for v_rot (...,3,3)."""
    lib, pre, _ = _lib(dtype)
    q = np.ascontiguousarray(np.asarray(quats, dtype).reshape(-1, 4))
    g = np.ascontiguousarray(np.asarray(v_rot, dtype).reshape(-1, 9))
    out = np.empty_like(q)
    getattr(lib, pre + "quat_to_rotmat_bwd")(C.c_int(q.shape[0]), _p(q), _p(g), _p(out))
    return out.reshape(np.shape(quats))


def mlp_fwd(x, w1, b1, w2, b2, dtype=np.float32):
    """Reference MLP(in, out, hidden_list=[128]) forward (gaussian_splatting.py:198-213):
    relu(x @ w1.T + b1) @ w2.T + b2, in the summation order of the HIP kernel."""
    lib, pre, _ = _lib(dtype)
    x = _c(x, dtype)
    shape = x.shape
    x = x.reshape(-1, shape[-1])
    w1, b1, w2, b2 = _c(w1, dtype), _c(b1, dtype), _c(w2, dtype), _c(b2, dtype)
    assert w1.shape == (128, x.shape[1]) and b1.shape == (128,) and w2.shape[1] == 128
    assert b2.shape == (w2.shape[0],)
    y = np.empty((x.shape[0], w2.shape[0]), dtype)
    getattr(lib, pre + "mlp_fwd")(C.c_int64(x.shape[0]), C.c_int(x.shape[1]), C.c_int(w2.shape[0]), _p(x),
                                  _p(w1), _p(b1), _p(w2), _p(b2), _p(y))
    return y.reshape(shape[:-1] + (w2.shape[0],))


# ------------------------------------------------------------------------------------------------
# densify / cull / Adam (pinned against torch in tests/test_densify_adam.py)
# ------------------------------------------------------------------------------------------------
def _u8(a):
    return np.ascontiguousarray(np.asarray(a).astype(np.uint8))


def adam_step(p, g, m, v, lr, beta1=0.9, beta2=0.999, eps=1e-8, weight_decay=0.0, step=1,
              dtype=np.float32):
    """One torch.optim.Adam step of one parameter array (reference engine/optimizers.py:158-171).
    Returns NEW (p, m, v); inputs are not modified."""
    lib, pre, _ = _lib(dtype)
    p, m, v = (_c(a, dtype).copy() for a in (p, m, v))
    g = _c(g, dtype)
    getattr(lib, pre + "adam_step")(C.c_int64(p.size), _p(p), _p(g), _p(m), _p(v), C.c_double(lr),
                                    C.c_double(beta1), C.c_double(beta2), C.c_double(eps),
                                    C.c_double(weight_decay), C.c_int64(step))
    return p, m, v


def mask_scan(mask, invert=False):
    lib, pre, _ = _lib(np.float32)
    mask = _u8(mask).reshape(-1)
    ranks = np.empty(mask.size, np.int32)
    fn = getattr(lib, pre + "mask_scan")
    fn.restype = C.c_int64
    total = fn(C.c_int(mask.size), _p(mask), C.c_int(int(invert)), _p(ranks))
    return ranks, int(total)


def compact_rows(deleted_mask, arr, dtype=np.float32):
    """arr[~deleted_mask] (reference cull_gaussians :497-502, remove_from_optim :341-342)."""
    lib, pre, _ = _lib(dtype)
    deleted_mask = _u8(deleted_mask).reshape(-1)
    a = _c(arr, dtype)
    n = a.shape[0]
    w = int(a.size // max(n, 1)) if n else 1
    dst = np.empty_like(a)
    fn = getattr(lib, pre + "compact_rows")
    fn.restype = C.c_int64
    k = fn(C.c_int(n), _p(deleted_mask), C.c_int(w), _p(a), _p(dst))
    return dst[:int(k)]


ROWS_COPY, ROWS_MEANS, ROWS_SCALES, ROWS_ZERO_NEW = 0, 1, 2, 3


def densify_rows(arr, kind, split_mask, dup_mask, nsamps, samples, size_fac, means, scales, quats,
                 dtype=np.float32):
    """torch.cat([old, split samples, dups]) of one array (reference :434-439,:504-546,:352-371)."""
    lib, pre, RT = _lib(dtype)
    split_mask, dup_mask = _u8(split_mask).reshape(-1), _u8(dup_mask).reshape(-1)
    a = _c(arr, dtype)
    n = a.shape[0]
    w = int(a.size // n)
    ns, nd = int(split_mask.sum()), int(dup_mask.sum())
    dst = np.empty((n + nsamps * ns + nd,) + a.shape[1:], dtype)
    samples = _c(samples, dtype).reshape(-1, 3)
    assert samples.shape[0] == nsamps * ns
    getattr(lib, pre + "densify_rows")(C.c_int(n), _p(split_mask), _p(dup_mask), C.c_int(nsamps),
                                       _p(samples), RT(size_fac), _p(_c(means, dtype)), _p(_c(scales, dtype)),
                                       _p(_c(quats, dtype)), C.c_int(w), C.c_int(kind), _p(a), _p(dst))
    return dst


def densify_stats(xys_grad, radii, max_dim, first, grad_norm, vis_counts, max_2dsize, dtype=np.float32):
    """after_train (:373-393); returns NEW (grad_norm, vis_counts, max_2dsize)."""
    lib, pre, _ = _lib(dtype)
    xg = _c(xys_grad, dtype)
    n = xg.shape[0]
    rad = np.ascontiguousarray(radii, np.int32)
    outs = [np.zeros(n, dtype) if a is None else _c(a, dtype).copy() for a in (grad_norm, vis_counts, max_2dsize)]
    getattr(lib, pre + "densify_stats")(C.c_int(n), _p(xg), _p(rad), C.c_int(max_dim), C.c_int(int(first)),
                                        _p(outs[0]), _p(outs[1]), _p(outs[2]))
    return tuple(outs)


def densify_masks(grad_norm, vis_counts, max_2dsize, scales, max_dim, grad_thresh, size_thresh,
                  split_screen_size, use_screen, dtype=np.float32, size_fac=1.6):
    lib, pre, RT = _lib(dtype)
    n = np.asarray(grad_norm).shape[0]
    sm, dm = np.empty(n, np.uint8), np.empty(n, np.uint8)
    getattr(lib, pre + "densify_masks")(C.c_int(n), _p(_c(grad_norm, dtype)), _p(_c(vis_counts, dtype)),
                                        _p(_c(max_2dsize, dtype)), _p(_c(scales, dtype)), C.c_int(max_dim),
                                        RT(grad_thresh), RT(size_thresh), RT(split_screen_size),
                                        C.c_int(int(use_screen)), RT(size_fac), _p(sm), _p(dm))
    return sm.astype(bool), dm.astype(bool)


def cull_mask(opacities, scales, max_2dsize, alpha_thresh, scale_thresh, screen_thresh, use_scale,
              use_screen, dtype=np.float32):
    lib, pre, RT = _lib(dtype)
    op = _c(opacities, dtype).reshape(-1)
    n = op.shape[0]
    m2 = _c(max_2dsize if max_2dsize is not None else np.zeros(n), dtype)
    out = np.empty(n, np.uint8)
    getattr(lib, pre + "cull_mask")(C.c_int(n), _p(op), _p(_c(scales, dtype)), _p(m2), RT(alpha_thresh),
                                    RT(scale_thresh), RT(screen_thresh), C.c_int(int(use_scale)),
                                    C.c_int(int(use_screen)), _p(out))
    return out.astype(bool)


# ------------------------------------------------------------------------------------------------
# SURVEY 8f-2 second half: MLP backward, cosine-similarity loss (pinned to torch autograd in tests)
# ------------------------------------------------------------------------------------------------
def mlp_bwd(x, w1, b1, w2, g, dtype=np.float32):
    """-> (v_x, v_w1, v_b1, v_w2, v_b2) of y = W2 relu(W1 x + b1) + b2 for g = dL/dy."""
    lib, pre, _ = _lib(dtype)
    x, g = _c(x, dtype), _c(g, dtype)
    shape = x.shape
    x, g = x.reshape(-1, shape[-1]), g.reshape(-1, g.shape[-1])
    w1, b1, w2 = _c(w1, dtype), _c(b1, dtype), _c(w2, dtype)
    v_x, v_w1, v_b1 = np.empty_like(x), np.empty_like(w1), np.empty_like(b1)
    v_w2, v_b2 = np.empty_like(w2), np.empty(w2.shape[0], dtype)
    getattr(lib, pre + "mlp_bwd")(C.c_int64(x.shape[0]), C.c_int(x.shape[1]), C.c_int(w2.shape[0]), _p(x), _p(w1),
                                  _p(b1), _p(w2), _p(g), _p(v_x), _p(v_w1), _p(v_b1), _p(v_w2), _p(v_b2))
    return v_x.reshape(shape), v_w1, v_b1, v_w2, v_b2


def cosine_loss_fwd(a, b, dtype=np.float32):
    """a, b (M,C) -> (loss, sim (M,), |a| (M,), |b| (M,)); reference cosine_similarity_loss :113-118."""
    lib, pre, RT = _lib(dtype)
    a, b = _c(a, dtype), _c(b, dtype)
    m, c = a.shape
    sim, na, nb = np.empty(m, dtype), np.empty(m, dtype), np.empty(m, dtype)
    fn = getattr(lib, pre + "cosine_loss_fwd")
    fn.restype = RT
    loss = fn(C.c_int64(m), C.c_int(c), _p(a), _p(b), _p(sim), _p(na), _p(nb))
    return float(loss), sim, na, nb


def cosine_loss_bwd(a, b, sim, na, nb, v_loss, dtype=np.float32):
    lib, pre, RT = _lib(dtype)
    a, b = _c(a, dtype), _c(b, dtype)
    m, c = a.shape
    v_a, v_b = np.empty_like(a), np.empty_like(b)
    getattr(lib, pre + "cosine_loss_bwd")(C.c_int64(m), C.c_int(c), _p(a), _p(b), _p(_c(sim, dtype)),
                                          _p(_c(na, dtype)), _p(_c(nb, dtype)), RT(v_loss), _p(v_a), _p(v_b))
    return v_a, v_b
