"""The "compat" variant: the places where gsplat 0.1.0 is RECALLED to deviate from exact calculus
(PARITY.md) and the least certain constant (GG_ALPHA_MAX_BWD, 0.999 vs 0.99), switched by macros in
include/gg_constants.h.  Both the oracle and the HIP library are also built with the switches on
(libgg_oracle_compat_*.so, libgg_raster_compat.so); this file holds the two variant builds to each
other exactly as the default builds are held to each other, so a later session that has the gsplat
source only flips constants."""
import os

import numpy as np
import pytest
import torch

from gaussiangrasper_amd.camera import ring_cameras
from gaussiangrasper_amd.scene import make_scene

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture()
def compat_oracle(oracle):
    oracle.use_variant("compat")
    yield oracle
    oracle.use_variant("")


def _project_case(n=4000, h=96, w=128, fov=60.0):
    sc = make_scene(n, config_index=12)
    sc.means.mul_(2.5)                                   # spread: many Gaussians beyond 1.3 tan(fov/2)
    v = ring_cameras(3, h, w, fov_x_deg=fov)[1]
    rng = np.random.default_rng(3)
    cot = (rng.standard_normal((n, 2)).astype(np.float32), rng.standard_normal(n).astype(np.float32),
           rng.standard_normal((n, 3)).astype(np.float32))
    return sc, v, cot


def _oracle_project_bwd(O, sc, v, cot, dtype=np.float32):
    a = (sc.means.numpy(), sc.scales.exp().numpy(), 1.0, sc.quats.numpy(), v.viewmat[:3].numpy(),
         v.projmat.numpy(), v.fx, v.fy, v.cx, v.cy, v.height, v.width)
    fwd = O.project_fwd(*a, v.tile_bounds, dtype=dtype)
    return fwd, O.project_bwd(*a, fwd[2], fwd[3], *cot, dtype=dtype)


def test_compat_oracle_differs_from_the_exact_vjp_only_where_documented(oracle):
    sc, v, cot = _project_case()
    fwd, exact = _oracle_project_bwd(oracle, sc, v, cot, np.float64)
    oracle.use_variant("compat")
    try:
        fwd_c, compat = _oracle_project_bwd(oracle, sc, v, cot, np.float64)
    finally:
        oracle.use_variant("")
    for a, b in zip(fwd, fwd_c):
        assert np.array_equal(a, b)                      # the forward is the same function
    vis = fwd[2] > 0
    assert vis.sum() > 500
    # scales: only the EWA path -> identical for Gaussians inside the FOV clamp, different beyond it
    mv = sc.means.numpy().astype(np.float64) @ v.viewmat[:3, :3].numpy().astype(np.float64).T \
        + v.viewmat[:3, 3].numpy().astype(np.float64)
    lim_x, lim_y = 1.3 * 0.5 * v.width / v.fx, 1.3 * 0.5 * v.height / v.fy
    clamped = vis & ((np.abs(mv[:, 0] / mv[:, 2]) > lim_x) | (np.abs(mv[:, 1] / mv[:, 2]) > lim_y))
    inside = vis & ~clamped
    assert np.allclose(exact[1][inside], compat[1][inside], rtol=1e-12, atol=0)
    if clamped.any():
        assert not np.allclose(exact[1][clamped], compat[1][clamped], rtol=1e-6)
    # means: the dropped w-path shows everywhere something is visible
    assert not np.allclose(exact[0][vis], compat[0][vis], rtol=1e-6)
    # quaternions: compat = gradient w.r.t. the normalised components; for unit quaternions the exact one
    # is its projection onto the tangent space, so the two agree after that projection
    q = sc.quats.numpy().astype(np.float64)
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    proj = compat[2] - q * np.sum(q * compat[2], axis=1, keepdims=True)
    assert np.allclose(proj[inside], exact[2][inside], rtol=1e-6, atol=1e-9)    # fp32-normalised inputs: |q| = 1 +- 6e-8


def test_alpha_max_bwd_only_matters_where_the_clamp_is_hit(oracle):
    """blend backward at GG_ALPHA_MAX_BWD 0.999 (default) and 0.99 (compat) on the golden-style inputs"""
    from test_gpu_parity import _blend_inputs
    n, h, w, ch = 400, 32, 32, 3
    xys, depths, radii, conics, nth, colors, opac, bg = _blend_inputs(oracle, n, h, w, ch, seed=9)
    opac[: n // 4] = 0.9995                              # some Gaussians above both clamps
    out, saved = oracle.rasterize_fwd(xys, depths, radii, conics, nth, colors, opac, h, w, bg)
    v_out = np.random.default_rng(1).standard_normal(out.shape).astype(np.float32)
    b = saved["bins"]
    args = (b["gaussian_ids_sorted"], b["tile_bins"], xys, conics, colors, opac, h, w, bg, saved["final_Ts"],
            saved["final_idx"], v_out)
    g0 = oracle.blend_bwd(*args)
    oracle.use_variant("compat")
    try:
        g1 = oracle.blend_bwd(*args)
    finally:
        oracle.use_variant("")
    assert any(not np.allclose(a, c, rtol=1e-5, atol=1e-7) for a, c in zip(g0, g1))
    opac2 = np.minimum(opac, 0.5)                        # nothing reaches either clamp: same gradients
    out2, saved2 = oracle.rasterize_fwd(xys, depths, radii, conics, nth, colors, opac2, h, w, bg)
    args2 = (saved2["bins"]["gaussian_ids_sorted"], saved2["bins"]["tile_bins"], xys, conics, colors, opac2, h, w,
             bg, saved2["final_Ts"], saved2["final_idx"], v_out)
    g0 = oracle.blend_bwd(*args2)
    oracle.use_variant("compat")
    try:
        g1 = oracle.blend_bwd(*args2)
    finally:
        oracle.use_variant("")
    for a, c in zip(g0, g1):
        assert np.array_equal(a, c)


DEV = "cuda:0"


@pytest.mark.gpu
def test_gpu_compat_library_matches_the_compat_oracle(compat_oracle):
    """libgg_raster_compat.so through the C ABI against libgg_oracle_compat_f32.so: project_bwd and
    blend_bwd with the tolerances of the default pair (tests/test_gpu_parity.py)"""
    from gaussiangrasper_amd import _lib, build as gg_build
    from gaussiangrasper_amd import ops as P
    from test_gpu_parity import _blend_inputs, assert_close
    gg_build.build_compat(force=False)        # rebuilt here if a source is newer than the variant library
    lib = _lib.load_variant(gg_build.COMPAT_OUT)
    O = compat_oracle
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(DEV)
    ptr = P._ptr
    # ---- project_bwd
    sc, v, cot = _project_case(20000, 300, 400)
    fwd, ref = _oracle_project_bwd(O, sc, v, cot)
    n = sc.num_points
    m, s, q = t(sc.means.numpy()), t(sc.scales.exp().numpy()), t(sc.quats.numpy())
    vm, pm = t(v.viewmat[:3].numpy()), t(v.projmat.numpy())
    outs = [torch.empty(n, k, device=DEV) for k in (3, 3, 4)]
    stream = P._stream(m.device)
    rad_t, con_t, cot_t = t(fwd[2]), t(fwd[3]), [t(c) for c in cot]     # (kept alive across the launches)
    st = lib.gg_project_bwd(n, ptr(m), ptr(s), 1.0, ptr(q), ptr(vm), ptr(pm), v.fx, v.fy, v.cx, v.cy, v.height,
                            v.width, ptr(rad_t), ptr(con_t), ptr(cot_t[0]), ptr(cot_t[1]),
                            ptr(cot_t[2]), ptr(outs[0]), ptr(outs[1]), ptr(outs[2]), stream)
    assert st == 0
    for name, got, r in zip(("v_mean3d", "v_scale", "v_quat"), outs, ref):
        assert_close(got.cpu().numpy(), r, f"compat.project_bwd.{name}", rtol=1e-6, atol_frac=1e-7)
    # and it is NOT the default library's answer
    dflt = _lib.load()
    outs_d = [torch.empty(n, k, device=DEV) for k in (3, 3, 4)]
    dflt.gg_project_bwd(n, ptr(m), ptr(s), 1.0, ptr(q), ptr(vm), ptr(pm), v.fx, v.fy, v.cx, v.cy, v.height, v.width,
                        ptr(rad_t), ptr(con_t), ptr(cot_t[0]), ptr(cot_t[1]), ptr(cot_t[2]),
                        ptr(outs_d[0]), ptr(outs_d[1]), ptr(outs_d[2]), stream)
    assert not torch.allclose(outs[0], outs_d[0], rtol=1e-4, atol=1e-6)
    # ---- blend_bwd at GG_ALPHA_MAX_BWD = 0.99
    for ch in (3, 32):
        n, h, w = 3000, 64, 80
        xys, depths, radii, conics, nth, colors, opac, bg = _blend_inputs(O, n, h, w, ch, seed=5)
        opac[: n // 5] = 0.9995
        ref_out, saved = O.rasterize_fwd(xys, depths, radii, conics, nth, colors, opac, h, w, bg)
        v_out = np.random.default_rng(4).standard_normal(ref_out.shape).astype(np.float32)
        b = saved["bins"]
        ref = O.blend_bwd(b["gaussian_ids_sorted"], b["tile_bins"], xys, conics, colors, opac, h, w, bg,
                          saved["final_Ts"], saved["final_idx"], v_out)
        ids, bins_t = t(b["gaussian_ids_sorted"].astype(np.int32)), t(b["tile_bins"].astype(np.int32))
        vx, vc, vo_ = (torch.empty(n, k, device=DEV) for k in (2, 3, 1))
        vcol = torch.empty(n, ch, device=DEV)
        ws = torch.empty(lib.gg_blend_workspace(n), dtype=torch.uint8, device=DEV)
        keep = [t(a) for a in (xys, conics, colors, opac, bg, saved["final_Ts"],
                               saved["final_idx"].astype(np.int32), v_out)]
        st = lib.gg_blend_bwd(ch, n, h, w, ptr(ids), ptr(bins_t), *[ptr(k) for k in keep], ptr(vx), ptr(vc),
                              ptr(vcol), ptr(vo_), 0, 0, ptr(ws), ws.numel(), 0, stream)
        assert st == 0
        for name, g, r in zip(("v_xy", "v_conic", "v_colors", "v_opacity"), (vx, vc, vcol, vo_), ref):
            assert_close(g.cpu().numpy(), r.reshape(g.shape), f"compat.blend_bwd<{ch}>.{name}", rtol=5e-5,
                         atol_frac=1e-6)
