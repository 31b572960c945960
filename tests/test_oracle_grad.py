"""Finite-difference checks (fp64 oracle build) of every analytic backward the oracle defines —
SURVEY.md §8c(2).  The f32 parity build and the HIP kernels use the same formulas; the GPU tests
compare against the f32 oracle, these tests tie those formulas to the forward."""
import numpy as np
import pytest
import torch

from gaussiangrasper_amd.camera import ring_cameras
from gaussiangrasper_amd.scene import make_scene

F64 = np.float64


def _setup(n=40, h=32, w=48, smul=14.0, seed=21):
    sc = make_scene(n, feature_dim=5, config_index=seed)
    v = ring_cameras(3, h, w)[1]
    means = sc.means.numpy().astype(F64) * 0.35          # keep everything well inside the frustum
    scales = sc.scales.exp().numpy().astype(F64) * smul
    quats = sc.quats.numpy().astype(F64) * 1.3           # un-normalised: exercises the q/|q| VJP
    opac = np.clip(torch.sigmoid(sc.opacities).numpy().astype(F64), 0.05, 0.95)
    return sc, v, means, scales, quats, opac


def _project(O, v, means, scales, quats):
    return O.project_fwd(means, scales, 1.0, quats, v.viewmat[:3].numpy(), v.projmat.numpy(), v.fx,
                         v.fy, v.cx, v.cy, v.height, v.width, v.tile_bounds, dtype=F64)


def _fd(f, x, idxs, eps):
    g = np.zeros(len(idxs))
    for k, ix in enumerate(idxs):
        xp, xm = x.copy(), x.copy()
        xp[ix] += eps
        xm[ix] -= eps
        g[k] = (f(xp) - f(xm)) / (2 * eps)
    return g


def _pick(rng, shape, count):
    flat = rng.choice(int(np.prod(shape)), size=min(count, int(np.prod(shape))), replace=False)
    return [tuple(int(t) for t in np.unravel_index(i, shape)) for i in flat]


def test_blend_bwd_matches_finite_differences(oracle):
    O = oracle
    sc, v, means, scales, quats, opac = _setup()
    xys, depths, radii, conics, nth, _ = _project(O, v, means, scales, quats)
    assert (radii > 0).sum() > 30
    h, w = v.height, v.width
    rng = np.random.default_rng(0)
    ch = 5
    colors = rng.uniform(-1, 1, (len(means), ch))
    bg = rng.uniform(0, 1, ch)
    v_out = rng.standard_normal((h, w, ch))
    b = O.bin_and_sort(xys, depths, radii, nth, v.tile_bounds, dtype=F64)

    def loss(xy=xys, con=conics, col=colors, op=opac):
        out, _, _ = O.blend_fwd(b["gaussian_ids_sorted"], b["tile_bins"], xy, con, col, op, h, w, bg, dtype=F64)
        return float((out * v_out).sum())

    out, ft, fi = O.blend_fwd(b["gaussian_ids_sorted"], b["tile_bins"], xys, conics, colors, opac, h, w, bg, dtype=F64)
    v_xy, v_conic, v_col, v_op = O.blend_bwd(b["gaussian_ids_sorted"], b["tile_bins"], xys, conics, colors,
                                             opac, h, w, bg, ft, fi, v_out, dtype=F64)
    vis = np.nonzero(radii > 0)[0]
    pick = lambda shape_cols: [(int(g), int(c)) for g in rng.choice(vis, 12) for c in range(shape_cols)]
    # colours: exact linear dependence
    idx = pick(ch)[:25]
    fd = _fd(lambda x: loss(col=x), colors, idx, 1e-5)
    assert np.allclose(fd, [v_col[i] for i in idx], rtol=1e-6, atol=1e-8)
    # opacity
    idx = [(int(g), 0) for g in rng.choice(vis, 20)]
    fd = _fd(lambda x: loss(op=x), opac, idx, 1e-6)
    assert np.allclose(fd, [v_op[i] for i in idx], rtol=2e-4, atol=1e-6)
    # xy
    idx = pick(2)[:24]
    fd = _fd(lambda x: loss(xy=x), xys, idx, 1e-6)
    assert np.allclose(fd, [v_xy[i] for i in idx], rtol=2e-4, atol=1e-6)
    # conic: gsplat convention — v_conic[:,1] is HALF the derivative w.r.t. conic.y
    idx = pick(3)[:30]
    fd = _fd(lambda x: loss(con=x), conics, idx, 1e-7)
    ana = np.array([v_conic[i] * (2.0 if i[1] == 1 else 1.0) for i in idx])
    assert np.allclose(fd, ana, rtol=5e-4, atol=1e-5)


def test_project_bwd_matches_finite_differences(oracle):
    O = oracle
    sc, v, means, scales, quats, opac = _setup(n=30)
    n = len(means)
    rng = np.random.default_rng(1)
    wx, wd, wc = rng.standard_normal((n, 2)), rng.standard_normal(n), rng.standard_normal((n, 3))

    def loss(m=means, s=scales, q=quats):
        xys, depths, radii, conics, nth, _ = _project(O, v, m, s, q)
        # symmetric-matrix convention for the conic cotangent: off-diagonal counted twice
        return float((xys * wx).sum() + (depths * wd).sum() + (conics * wc * np.array([1, 2, 1])).sum())

    xys, depths, radii, conics, nth, _ = _project(O, v, means, scales, quats)
    assert (radii > 0).all()
    vm, vs, vq = O.project_bwd(means, scales, 1.0, quats, v.viewmat[:3].numpy(), v.projmat.numpy(), v.fx,
                               v.fy, v.cx, v.cy, v.height, v.width, radii, conics, wx, wd, wc, dtype=F64)
    for arr, ana, name in ((means, vm, "means"), (scales, vs, "scales"), (quats, vq, "quats")):
        idx = _pick(rng, arr.shape, 30)
        kw = {"means": "m", "scales": "s", "quats": "q"}[name]
        fd = _fd(lambda x: loss(**{kw: x}), arr, idx, 1e-6)
        a = np.array([ana[i] for i in idx])
        assert np.allclose(fd, a, rtol=1e-4, atol=1e-5 * np.abs(a).max()), name
    # radial component of v_quat vanishes (projection through q/|q|)
    assert np.abs((vq * quats).sum(-1)).max() < 1e-8 * np.abs(vq).max()


def test_project_bwd_fov_clamp_branch(oracle):
    """a Gaussian beyond 1.3*tan(fov/2): the EWA Jacobian uses the clamped t; check that branch's VJP"""
    O = oracle
    v = ring_cameras(3, 32, 48)[0]
    cam = v.cam_pos.numpy().astype(F64)
    fwd = -cam / np.linalg.norm(cam)
    right = np.cross(fwd, [0, 0, 1.0])
    right /= np.linalg.norm(right)
    lim = 1.3 * 0.5 * v.width / v.fx
    means = np.array([cam + 2.0 * fwd + 2.0 * (lim + 0.15) * right])   # t.x/t.z just past the limit
    scales = np.full((1, 3), 0.9)
    quats = np.array([[0.9, 0.1, -0.2, 0.3]])
    xys, depths, radii, conics, nth, _ = _project(O, v, means, scales, quats)
    assert radii[0] > 0, "needs to stay visible (large scale) so the gradient is live"
    wc = np.array([[0.7, -0.4, 1.1]])

    def loss(m):
        return float((_project(O, v, m, scales, quats)[3] * wc * np.array([1, 2, 1])).sum())
    vm, _, _ = O.project_bwd(means, scales, 1.0, quats, v.viewmat[:3].numpy(), v.projmat.numpy(), v.fx,
                             v.fy, v.cx, v.cy, v.height, v.width, radii, conics, np.zeros((1, 2)),
                             np.zeros(1), wc, dtype=F64)
    fd = _fd(loss, means, [(0, 0), (0, 1), (0, 2)], 1e-6)
    assert np.allclose(fd, vm[0], rtol=1e-4, atol=1e-9)


def test_sh_bwd_matches_finite_differences(oracle):
    O = oracle
    rng = np.random.default_rng(2)
    n = 6
    d = rng.standard_normal((n, 3))
    cf = rng.standard_normal((n, 25, 3))
    vc = rng.standard_normal((n, 3))
    for deg in (0, 1, 2, 3, 4):
        ana = O.sh_bwd(deg, 25, d, vc, dtype=F64)
        idx = _pick(rng, cf.shape, 40)
        fd = _fd(lambda x: float((O.sh_fwd(deg, d, x, dtype=F64) * vc).sum()), cf, idx, 1e-5)
        assert np.allclose(fd, [ana[i] for i in idx], rtol=1e-6, atol=1e-9), deg
        nb = [1, 4, 9, 16, 25][deg]
        assert not ana[:, nb:, :].any()


def test_end_to_end_chain_matches_finite_differences(oracle):
    """means -> project -> blend -> loss: the composed analytic VJPs (incl. the half-gradient conic
    convention between blend_bwd and project_bwd) equal finite differences of the whole forward"""
    O = oracle
    sc, v, means, scales, quats, opac = _setup(n=25, smul=16.0)
    h, w = v.height, v.width
    rng = np.random.default_rng(3)
    colors = rng.uniform(0, 1, (len(means), 3))
    bg = np.zeros(3)
    v_out = rng.standard_normal((h, w, 3))
    xys, depths, radii, conics, nth, _ = _project(O, v, means, scales, quats)
    b = O.bin_and_sort(xys, depths, radii, nth, v.tile_bounds, dtype=F64)
    ids, bins = b["gaussian_ids_sorted"], b["tile_bins"]

    def loss(m=means, s=scales, q=quats):
        xy, _, _, con, _, _ = _project(O, v, m, s, q)
        out, _, _ = O.blend_fwd(ids, bins, xy, con, colors, opac, h, w, bg, dtype=F64)  # lists held fixed
        return float((out * v_out).sum())

    out, ft, fi = O.blend_fwd(ids, bins, xys, conics, colors, opac, h, w, bg, dtype=F64)
    v_xy, v_conic, _, _ = O.blend_bwd(ids, bins, xys, conics, colors, opac, h, w, bg, ft, fi, v_out, dtype=F64)
    vm, vs, vq = O.project_bwd(means, scales, 1.0, quats, v.viewmat[:3].numpy(), v.projmat.numpy(), v.fx,
                               v.fy, v.cx, v.cy, h, w, radii, conics, v_xy, np.zeros(len(means)), v_conic, dtype=F64)
    for arr, ana, kw in ((means, vm, "m"), (scales, vs, "s"), (quats, vq, "q")):
        idx = _pick(rng, arr.shape, 24)
        fd = _fd(lambda x: loss(**{kw: x}), arr, idx, 1e-6)
        a = np.array([ana[i] for i in idx])
        assert np.allclose(fd, a, rtol=5e-4, atol=2e-5 * np.abs(ana).max()), kw


def test_f32_oracle_tracks_f64_oracle(oracle):
    """same formulas, two precisions: the parity build stays within fp32 rounding of the fp64 build"""
    O = oracle
    sc, v, means, scales, quats, opac = _setup(n=60)
    a = _project(O, v, means, scales, quats)
    b = O.project_fwd(means, scales, 1.0, quats, v.viewmat[:3].numpy(), v.projmat.numpy(), v.fx, v.fy,
                      v.cx, v.cy, v.height, v.width, v.tile_bounds)
    assert np.array_equal(a[2], b[2]) and np.array_equal(a[4], b[4])
    for x, y in zip((a[0], a[1], a[3], a[5]), (b[0], b[1], b[3], b[5])):
        assert np.allclose(x, y, rtol=3e-5, atol=1e-6)
