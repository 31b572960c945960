"""TEST-ONLY: gsplat-style autograd.Functions backed by the CPU oracle, so the reference's call
sequence (gaussiangrasper_amd.pipeline.render_view) can be replayed on CPU tensors — BASELINE
config 1 "plumbing, no GPU" — and compared with the HIP operators on the GPU box.
The product never imports this module."""
import numpy as np
import torch
from torch.autograd import Function

from oracle import oracle as O


def _np(t):
    return t.detach().cpu().numpy()


class ProjectGaussians(Function):
    @staticmethod
    def forward(ctx, means3d, scales, glob_scale, quats, viewmat, projmat, fx, fy, cx, cy,
                img_height, img_width, tile_bounds, clip_thresh=0.01):
        args = (_np(means3d), _np(scales), float(glob_scale), _np(quats), _np(viewmat),
                _np(projmat), fx, fy, cx, cy, img_height, img_width)
        xys, depths, radii, conics, nth, cov3d = O.project_fwd(*args, tile_bounds, clip_thresh)
        ctx.args = args
        ctx.radii, ctx.conics = radii, conics
        outs = [torch.from_numpy(a) for a in (xys, depths, radii, conics, nth, cov3d)]
        ctx.mark_non_differentiable(outs[2], outs[4])
        return tuple(outs)

    @staticmethod
    def backward(ctx, v_xys, v_depths, v_radii, v_conics, v_nth, v_cov3d):
        n = ctx.args[0].shape[0]
        z = lambda t, s: np.zeros(s, np.float32) if t is None else _np(t)
        vm, vs, vq = O.project_bwd(*ctx.args, ctx.radii, ctx.conics, z(v_xys, (n, 2)),
                                   z(v_depths, (n,)), z(v_conics, (n, 3)))
        return (torch.from_numpy(vm), torch.from_numpy(vs), None, torch.from_numpy(vq)) + (None,) * 10


class SphericalHarmonics(Function):
    @staticmethod
    def forward(ctx, degrees_to_use, viewdirs, coeffs):
        ctx.deg, ctx.k, ctx.vd = degrees_to_use, coeffs.shape[-2], _np(viewdirs)
        return torch.from_numpy(O.sh_fwd(degrees_to_use, ctx.vd, _np(coeffs)))

    @staticmethod
    def backward(ctx, v_colors):
        return None, None, torch.from_numpy(O.sh_bwd(ctx.deg, ctx.k, ctx.vd, _np(v_colors)))


class _Rasterize(Function):
    @staticmethod
    def forward(ctx, xys, depths, radii, conics, num_tiles_hit, colors, opacity, img_height,
                img_width, background=None):
        if background is None:
            background = torch.ones(colors.shape[-1])
        a = dict(xys=_np(xys), conics=_np(conics), colors=_np(colors), opacity=_np(opacity),
                 bg=_np(background), h=int(img_height), w=int(img_width))
        out, saved = O.rasterize_fwd(a["xys"], _np(depths), _np(radii), a["conics"],
                                     _np(num_tiles_hit), a["colors"], a["opacity"], a["h"], a["w"],
                                     a["bg"])
        ctx.a, ctx.saved, ctx.oshape = a, saved, tuple(opacity.shape)
        return torch.from_numpy(out)

    @staticmethod
    def backward(ctx, v_out):
        a, s = ctx.a, ctx.saved
        n, ch = a["colors"].shape
        if s["bins"]["num_intersects"] < 1:
            g = (np.zeros((n, 2), np.float32), np.zeros((n, 3), np.float32),
                 np.zeros((n, ch), np.float32), np.zeros((n, 1), np.float32))
        else:
            b = s["bins"]
            g = O.blend_bwd(b["gaussian_ids_sorted"], b["tile_bins"], a["xys"], a["conics"],
                            a["colors"], a["opacity"], a["h"], a["w"], a["bg"], s["final_Ts"],
                            s["final_idx"], _np(v_out))
        v_xy, v_conic, v_colors, v_opacity = [torch.from_numpy(x) for x in g]
        return (v_xy, None, None, v_conic, None, v_colors, v_opacity.reshape(ctx.oshape), None,
                None, None)


class RasterizeGaussians(_Rasterize):
    pass


class NDRasterizeGaussians(_Rasterize):
    pass


class _QuatToRotmat(Function):
    @staticmethod
    def forward(ctx, quat):
        ctx.q = _np(quat)
        return torch.from_numpy(O.quat_to_rotmat(ctx.q))

    @staticmethod
    def backward(ctx, v_rot):
        return torch.from_numpy(O.quat_to_rotmat_bwd(ctx.q, _np(v_rot)))


def quat_to_rotmat(quat):
    return _QuatToRotmat.apply(quat)


def quat_to_rotmat_torch(quat):
    """The published torch expression of gsplat._torch_impl.quat_to_rotmat (†) — the plain torch
    reference the oracle's and the HIP kernel's forward and backward are checked against."""
    w, x, y, z = torch.unbind(torch.nn.functional.normalize(quat, dim=-1), dim=-1)
    mat = torch.stack([
        1 - 2 * (y ** 2 + z ** 2), 2 * (x * y - w * z), 2 * (x * z + w * y),
        2 * (x * y + w * z), 1 - 2 * (x ** 2 + z ** 2), 2 * (y * z - w * x),
        2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x ** 2 + y ** 2)], dim=-1)
    return mat.reshape(quat.shape[:-1] + (3, 3))
