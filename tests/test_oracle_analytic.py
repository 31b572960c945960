"""Self-pinning of the CPU oracle (the reference offers no fixture, so the oracle is
pinned by closed-form cases): projection of known configurations, single-Gaussian alpha map,
two-Gaussian compositing, culls, border clamps, the T<=1e-4 stop, transmittance conservation,
sort order, and the committed golden vectors."""
import glob
import math
import os

import numpy as np
import pytest
import torch

from gaussiangrasper_amd.camera import projection_matrix, ring_cameras, view_from_c2w
from gaussiangrasper_amd.scene import make_scene


def _front_view(h=64, w=96, f=80.0):
    """camera at the origin looking down world +z after the reference's flip: OpenGL c2w = identity
    looks down -z, so place Gaussians at negative world z."""
    return view_from_c2w(torch.eye(4), f, f, w / 2.0, h / 2.0, h, w)


def test_expf_against_libm(oracle):
    x = np.concatenate([np.linspace(-80, 0, 100001), -np.logspace(-10, 1.9, 2000)]).astype(np.float32)
    y = oracle.expf(x)
    ref = np.exp(x.astype(np.float64))
    rel = np.abs(y - ref) / ref
    assert rel.max() < 2.5e-7, rel.max()
    assert oracle.expf(np.array([0.0], np.float32))[0] == 1.0
    assert oracle.expf(np.array([-80.5, -1e9, -np.inf], np.float32)).tolist() == [0.0, 0.0, 0.0]
    # fp64 build is plain exp
    assert np.allclose(oracle.expf(x[:100], np.float64), np.exp(x[:100].astype(np.float64)))


def test_projection_matrix_and_view():
    p = projection_matrix(0.001, 1000, 2 * math.atan(0.5), 2 * math.atan(0.25))
    assert p.shape == (4, 4)
    assert math.isclose(p[0, 0].item(), 2.0, rel_tol=1e-6) and math.isclose(p[1, 1].item(), 4.0, rel_tol=1e-6)
    assert p[3].tolist() == [0.0, 0.0, 1.0, 0.0]   # w_clip = z_view
    v = ring_cameras(4, 30, 40)[1]
    w2c = v.viewmat
    assert torch.allclose(w2c[:3, :3] @ w2c[:3, :3].T, torch.eye(3), atol=1e-6)
    assert torch.allclose((w2c @ torch.cat([v.cam_pos, torch.ones(1)]))[:3], torch.zeros(3), atol=1e-5)
    origin_cam = w2c @ torch.tensor([0, 0, 0, 1.0])
    assert origin_cam[2] > 2.4 and abs(origin_cam[0]) < 1e-5 and abs(origin_cam[1]) < 1e-5


def test_project_isotropic_on_axis(oracle):
    """isotropic Gaussian on the optical axis: closed-form xy, depth, cov2d, conic, radius, tiles"""
    v = _front_view()
    s, z = 0.05, 2.0
    means = np.array([[0.0, 0.0, -z]], np.float32)           # in front after the x-pi flip
    out = oracle.project_fwd(means, np.full((1, 3), s, np.float32), 1.0, np.array([[1, 0, 0, 0]], np.float32),
                             v.viewmat[:3].numpy(), v.projmat.numpy(), v.fx, v.fy, v.cx, v.cy,
                             v.height, v.width, v.tile_bounds)
    xys, depths, radii, conics, nth, cov3d = out
    assert np.allclose(depths, [z], rtol=1e-6)
    assert np.allclose(xys, [[v.cx - 0.5, v.cy - 0.5]], atol=1e-3)   # w_eps shifts it by ~1e-5
    var = (v.fx * s / z) ** 2 + 0.3
    assert np.allclose(conics, [[1 / var, 0.0, 1 / var]], rtol=1e-5, atol=1e-7)
    assert radii[0] == math.ceil(3 * math.sqrt(var))
    assert np.allclose(cov3d, [[s * s, 0, 0, s * s, 0, s * s]], rtol=1e-6)
    r = radii[0]
    x0, x1 = int((xys[0, 0] - r) / 16), int((xys[0, 0] + r) / 16 + 1)
    y0, y1 = int((xys[0, 1] - r) / 16), int((xys[0, 1] + r) / 16 + 1)
    assert nth[0] == (x1 - x0) * (y1 - y0)


def test_project_culls(oracle):
    v = _front_view()
    q = np.array([[1, 0, 0, 0]], np.float32)
    sc = np.full((1, 3), 0.02, np.float32)

    def run(p, scale=sc, clip=0.01):
        return oracle.project_fwd(np.array([p], np.float32), scale, 1.0, q, v.viewmat[:3].numpy(),
                                  v.projmat.numpy(), v.fx, v.fy, v.cx, v.cy, v.height, v.width,
                                  v.tile_bounds, clip)
    assert run([0, 0, +1.0])[2][0] == 0                      # behind the camera
    assert run([0, 0, -0.01])[2][0] == 0                     # z <= clip_thresh (equality culls)
    assert run([0, 0, -0.0101])[2][0] > 0
    assert run([0, 0, -0.5], clip=0.6)[2][0] == 0            # clip_thresh argument honoured
    far_off = run([50.0, 0, -2.0])                           # projects far outside the image
    assert far_off[2][0] == 0 and far_off[4][0] == 0 and far_off[0].tolist() == [[0.0, 0.0]]
    assert far_off[3].any()                                  # conic is written before the bbox cull
    # border: centre just outside the image but radius reaches in -> clamped bbox, still visible
    edge = run([-(v.cx + 1.0) * 2.0 / v.fx, 0, -2.0], scale=np.full((1, 3), 0.2, np.float32))
    assert edge[2][0] > 0 and 0 < edge[4][0] <= v.tile_bounds[0] * v.tile_bounds[1]


def test_quaternion_is_normalised_inside(oracle):
    v = _front_view()
    sc = make_scene(64, config_index=9)
    a = oracle.project_fwd(sc.means.numpy() * 0.3 - [0, 0, 2], sc.scales.exp().numpy() * 10, 1.0,
                           sc.quats.numpy(), v.viewmat[:3].numpy(), v.projmat.numpy(), v.fx, v.fy,
                           v.cx, v.cy, v.height, v.width, v.tile_bounds)
    b = oracle.project_fwd(sc.means.numpy() * 0.3 - [0, 0, 2], sc.scales.exp().numpy() * 10, 1.0,
                           sc.quats.numpy() * 3.7, v.viewmat[:3].numpy(), v.projmat.numpy(), v.fx,
                           v.fy, v.cx, v.cy, v.height, v.width, v.tile_bounds)
    assert np.array_equal(a[2], b[2]) and np.allclose(a[3], b[3], rtol=1e-4, atol=1e-7)


def _one_tile_inputs(xy, conic, n=None):
    xy = np.asarray(xy, np.float32).reshape(-1, 2)
    n = xy.shape[0]
    conic = np.asarray(conic, np.float32).reshape(n, 3)
    ids = np.arange(n, dtype=np.int32)
    bins = np.array([[0, n]], np.int32)
    return ids, bins, xy, conic


def test_single_gaussian_alpha_map(oracle):
    """one isotropic Gaussian: out = alpha*c + (1-alpha)*bg with alpha = min(.999, o*exp(-r^2/2var)),
    zero where alpha < 1/255"""
    var, o = 6.0, 0.8
    ids, bins, xy, conic = _one_tile_inputs([[7.3, 8.6]], [[1 / var, 0, 1 / var]])
    col, bg = np.array([[0.2, 0.5, 0.9]], np.float32), np.array([0.1, 0.0, 0.3], np.float32)
    out, ft, fi = oracle.blend_fwd(ids, bins, xy, conic, col, np.array([[o]], np.float32), 16, 16, bg)
    jj, ii = np.meshgrid(np.arange(16), np.arange(16))
    r2 = (7.3 - jj) ** 2 + (8.6 - ii) ** 2
    alpha = np.minimum(0.999, o * np.exp(-0.5 * r2 / var))
    alpha[alpha < 1 / 255] = 0
    expect = alpha[..., None] * col[0] + (1 - alpha[..., None]) * bg
    assert np.allclose(out, expect, atol=2e-6)
    assert np.allclose(ft, 1 - alpha, atol=2e-6)
    assert np.array_equal(fi, (alpha > 0).astype(np.int32))   # one past the last blended entry


def test_two_gaussians_composite_in_list_order(oracle):
    var = 4.0
    ids, bins, xy, conic = _one_tile_inputs([[8, 8], [8, 8]], [[1 / var, 0, 1 / var]] * 2)
    col = np.array([[1, 0, 0], [0, 1, 0]], np.float32)
    op = np.array([[0.5], [0.7]], np.float32)
    out, ft, fi = oracle.blend_fwd(ids, bins, xy, conic, col, op, 16, 16, np.zeros(3, np.float32))
    assert np.allclose(out[8, 8], [0.5, 0.5 * 0.7, 0], atol=1e-6) and np.isclose(ft[8, 8], 0.5 * 0.3, atol=1e-6)
    out2, _, _ = oracle.blend_fwd(ids[::-1].copy(), bins, xy, conic, col, op, 16, 16, np.zeros(3, np.float32))
    assert np.allclose(out2[8, 8], [0.3 * 0.5, 0.7, 0], atol=1e-6)


def test_alpha_clamp_and_early_stop(oracle):
    """opaque Gaussians: alpha clamps at 0.999; T goes 1 -> 1e-3 -> 1e-6<=1e-4: the second one is NOT
    blended, the pixel stops, final_idx stays 1"""
    ids, bins, xy, conic = _one_tile_inputs([[8, 8]] * 3, [[0.01, 0, 0.01]] * 3)
    col = np.array([[1, 0, 0], [0, 1, 0], [0, 0, 1]], np.float32)
    op = np.full((3, 1), 5.0, np.float32)
    out, ft, fi = oracle.blend_fwd(ids, bins, xy, conic, col, op, 16, 16, np.ones(3, np.float32))
    assert np.isclose(ft[8, 8], 1e-3, rtol=1e-3) and fi[8, 8] == 1
    assert np.allclose(out[8, 8], [0.999 + 1e-3, 1e-3, 1e-3], atol=1e-6)


def test_negative_sigma_is_skipped(oracle):
    """indefinite conic -> sigma < 0 on some pixels: those are skipped, not blended"""
    ids, bins, xy, conic = _one_tile_inputs([[8, 8]], [[0.1, 0.5, 0.1]])
    out, ft, fi = oracle.blend_fwd(ids, bins, xy, conic, np.ones((1, 3), np.float32),
                                   np.full((1, 1), 0.9, np.float32), 16, 16, np.zeros(3, np.float32))
    assert ft[9, 7] == 1.0 and out[9, 7].tolist() == [0, 0, 0]     # dx*dy < 0 -> sigma < 0
    assert ft[8, 8] < 0.2


def test_conservation_and_ragged_image(oracle):
    """sum_i alpha_i T_i + T_final = 1 on an image whose size is not a multiple of the tile"""
    h, w, n = 45, 70, 400
    sc = make_scene(n, config_index=4)
    v = ring_cameras(2, h, w)[0]
    xys, depths, radii, conics, nth, _ = oracle.project_fwd(
        sc.means.numpy(), sc.scales.exp().numpy() * 8, 1.0, sc.quats.numpy(), v.viewmat[:3].numpy(),
        v.projmat.numpy(), v.fx, v.fy, v.cx, v.cy, h, w, v.tile_bounds)
    op = torch.sigmoid(sc.opacities).numpy()
    ones, s = oracle.rasterize_fwd(xys, depths, radii, conics, nth, np.ones((n, 1), np.float32), op,
                                   h, w, np.zeros(1, np.float32))
    assert ones.shape == (h, w, 1)
    assert np.abs(ones[..., 0] + s["final_Ts"] - 1).max() < 2e-6
    assert (s["final_Ts"] < 0.9).mean() > 0.05


def test_binning_order_and_bins(oracle):
    h, w, n = 96, 128, 3000
    sc = make_scene(n, config_index=5)
    v = ring_cameras(2, h, w)[1]
    xys, depths, radii, conics, nth, _ = oracle.project_fwd(
        sc.means.numpy(), sc.scales.exp().numpy() * 6, 1.0, sc.quats.numpy(), v.viewmat[:3].numpy(),
        v.projmat.numpy(), v.fx, v.fy, v.cx, v.cy, h, w, v.tile_bounds)
    depths[::7] = depths[0]                       # force depth ties
    b = oracle.bin_and_sort(xys, depths, radii, nth, v.tile_bounds)
    I = b["num_intersects"]
    assert I == nth.sum() == b["cum_tiles_hit"][-1]
    ks, ids = b["isect_ids_sorted"], b["gaussian_ids_sorted"]
    assert (np.diff(ks) >= 0).all()
    tie = np.diff(ks) == 0
    assert tie.any() and (np.diff(ids)[tie] > 0).all()        # ties: ascending Gaussian id
    assert np.array_equal((ks & 0xFFFFFFFF).astype(np.uint32).view(np.float32), depths[ids])
    tiles = (ks >> 32).astype(np.int64)
    for t in range(v.tile_bounds[0] * v.tile_bounds[1]):
        lo, hi = b["tile_bins"][t]
        idx = np.nonzero(tiles == t)[0]
        if idx.size:
            assert (lo, hi) == (idx[0], idx[-1] + 1)
        else:
            assert (lo, hi) == (0, 0)
    # unsorted emission: row-major inside each bbox, contiguous per Gaussian
    g0 = int(np.argmax(nth))
    start = 0 if g0 == 0 else b["cum_tiles_hit"][g0 - 1]
    seg = (b["isect_ids"][start:start + nth[g0]] >> 32)
    assert (np.diff(seg) > 0).all() and (b["gaussian_ids"][start:start + nth[g0]] == g0).all()


def test_no_intersections_returns_background(oracle):
    n = 5
    z = np.zeros
    out, saved = oracle.rasterize_fwd(z((n, 2), np.float32), z(n, np.float32), z(n, np.int32),
                                      np.ones((n, 3), np.float32), z(n, np.int32),
                                      np.ones((n, 3), np.float32), np.ones((n, 1), np.float32), 20, 30,
                                      np.array([0.1, 0.2, 0.3], np.float32))
    assert saved["bins"]["num_intersects"] == 0
    assert np.allclose(out, np.broadcast_to(np.array([0.1, 0.2, 0.3], np.float32), (20, 30, 3)))


def test_quat_to_rotmat(oracle):
    """oracle forward/backward of quat_to_rotmat vs the published torch expression (autograd)."""
    import oracle_ops
    g = torch.Generator().manual_seed(0)
    q = (torch.randn(50, 4, generator=g) * 3).requires_grad_(True)     # deliberately unnormalised
    R = oracle_ops.quat_to_rotmat_torch(q)
    assert torch.allclose(R @ R.transpose(1, 2), torch.eye(3).expand(50, 3, 3), atol=1e-5)
    assert torch.allclose(torch.linalg.det(R), torch.ones(50), atol=1e-5)
    assert np.allclose(R.detach().numpy(), oracle.quat_to_rotmat(q.detach().numpy()), atol=1e-6)
    v = torch.randn(50, 3, 3, generator=g)
    (vq_t,) = torch.autograd.grad(R, q, v)
    vq_o = oracle.quat_to_rotmat_bwd(q.detach().numpy(), v.numpy())
    assert np.allclose(vq_o, vq_t.numpy(), atol=2e-6, rtol=1e-5)
    # float64 build against float64 autograd: tight
    q64 = q.detach().double().requires_grad_(True)
    (vq_t64,) = torch.autograd.grad(oracle_ops.quat_to_rotmat_torch(q64), q64, v.double())
    vq_o64 = oracle.quat_to_rotmat_bwd(q64.detach().numpy(), v.double().numpy(), dtype=np.float64)
    assert np.allclose(vq_o64, vq_t64.numpy(), atol=1e-13, rtol=1e-12)
    # the oracle-backed autograd.Function used by the CPU replays
    q2 = q.detach().clone().requires_grad_(True)
    (vq_f,) = torch.autograd.grad(oracle_ops.quat_to_rotmat(q2), q2, v)
    assert np.array_equal(vq_f.numpy(), vq_o)
    # batched leading dims
    assert oracle.quat_to_rotmat(np.ones((2, 5, 4), np.float32)).shape == (2, 5, 3, 3)
    # wxyz convention: 90 degrees about z maps x -> y
    s = math.sqrt(0.5)
    Rz = torch.from_numpy(oracle.quat_to_rotmat(np.array([[s, 0, 0, s]], np.float32)))[0]
    assert torch.allclose(Rz @ torch.tensor([1.0, 0, 0]), torch.tensor([0.0, 1, 0]), atol=1e-6)


def test_sh_band0_and_signs(oracle):
    """degree 0: colour = C0*coeff; degree 1 uses the 3DGS signs (-y, +z, -x)"""
    cf = np.zeros((3, 25, 3), np.float32)
    cf[:, 0, :] = 1.0
    cf[0, 1, 0] = 1.0   # multiplies -C1*y
    cf[1, 2, 1] = 1.0   # multiplies +C1*z
    cf[2, 3, 2] = 1.0   # multiplies -C1*x
    d = np.array([[0, 2.0, 0], [0, 0, 3.0], [4.0, 0, 0]], np.float32)   # un-normalised on purpose
    c0 = oracle.sh_fwd(0, d, cf)
    assert np.allclose(c0, 0.28209479177387814)
    c1 = oracle.sh_fwd(1, d, cf)
    C0, C1 = 0.28209479177387814, 0.4886025119029199
    assert np.isclose(c1[0, 0], C0 - C1) and np.isclose(c1[1, 1], C0 + C1) and np.isclose(c1[2, 2], C0 - C1)
    # higher bands are ignored below their degree
    cf[:, 9:, :] = 7.0
    assert np.allclose(oracle.sh_fwd(2, d, cf), oracle.sh_fwd(2, d, np.where(np.arange(25)[None, :, None] >= 9, 0, cf)))


def test_sh_basis_is_orthonormal(oracle):
    """Monte-Carlo orthonormality of the 25 basis functions the oracle evaluates (4pi * E[Yi Yj] = delta)"""
    rng = np.random.default_rng(0)
    d = rng.standard_normal((200000, 3))
    Y = oracle.sh_bwd(4, 25, d, np.ones((d.shape[0], 3)), dtype=np.float64)[:, :, 0]
    gram = 4 * np.pi * (Y.T @ Y) / d.shape[0]
    assert np.abs(gram - np.eye(25)).max() < 0.03


def test_golden_vectors(oracle):
    """the oracle reproduces its committed fixtures bit for bit (regression pin; tests/golden/make_golden.py)"""
    files = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "*.npz")))
    assert len(files) >= 5
    for f in files:
        z = np.load(f)
        h, w = int(z["hw"][0]), int(z["hw"][1])
        tb = ((w + 15) // 16, (h + 15) // 16, 1)
        fx, fy, cx, cy = [float(t) for t in z["intr"]]
        got = oracle.project_fwd(z["means"], z["scales"], 1.0, z["quats"], z["viewmat"], z["projmat"],
                                 fx, fy, cx, cy, h, w, tb)
        for name, g in zip(("xys", "depths", "radii", "conics", "num_tiles_hit", "cov3d"), got):
            assert np.array_equal(g, z[name]), (f, name)
        b = oracle.bin_and_sort(got[0], got[1], got[2], got[4], tb)
        assert np.array_equal(b["gaussian_ids_sorted"], z["gaussian_ids_sorted"])
        assert np.array_equal(b["isect_ids_sorted"], z["isect_ids_sorted"])
        assert np.array_equal(b["tile_bins"], z["tile_bins"])
        out, ft, fi = oracle.blend_fwd(b["gaussian_ids_sorted"], b["tile_bins"], got[0], got[3],
                                       z["colors"], z["opacity"], h, w, z["background"])
        assert np.array_equal(out, z["out_img"]) and np.array_equal(ft, z["final_Ts"]) and np.array_equal(fi, z["final_idx"])
        g = oracle.blend_bwd(b["gaussian_ids_sorted"], b["tile_bins"], got[0], got[3], z["colors"],
                             z["opacity"], h, w, z["background"], ft, fi, z["v_out"])
        for name, a in zip(("v_xy", "v_conic", "v_colors", "v_opacity"), g):
            assert np.array_equal(a, z[name]), (f, name)
        pm = oracle.project_bwd(z["means"], z["scales"], 1.0, z["quats"], z["viewmat"], z["projmat"],
                                fx, fy, cx, cy, h, w, got[2], got[3], z["v_xy"], np.zeros(len(got[2]), np.float32), z["v_conic"])
        for name, a in zip(("v_mean3d", "v_scale", "v_quat"), pm):
            assert np.array_equal(a, z[name]), (f, name)


def test_mlp_oracle_matches_the_published_module(oracle):
    """oracle.mlp_fwd against torch's Sequential(Linear(in,128), ReLU, Linear(128,out)) — the
    reference's MLP (gaussian_splatting.py:198-213).  fp32: 1e-5 (summation order differs);
    fp64 build: 1e-12."""
    g = torch.Generator().manual_seed(3)
    for in_dim, out_dim, rows in ((32, 512, 300), (8, 64, 37), (64, 96, 65)):
        seq = torch.nn.Sequential(torch.nn.Linear(in_dim, 128), torch.nn.ReLU(), torch.nn.Linear(128, out_dim))
        with torch.no_grad():
            for prm in seq.parameters():
                prm.copy_(torch.randn(prm.shape, generator=g) * 0.3)
        x = torch.randn(rows, in_dim, generator=g)
        w = [seq[0].weight, seq[0].bias, seq[2].weight, seq[2].bias]
        with torch.no_grad():
            ref32 = seq(x).numpy()
            ref64 = seq.double()(x.double()).numpy()
        got32 = oracle.mlp_fwd(x.numpy(), *[q.detach().float().numpy() for q in w])
        got64 = oracle.mlp_fwd(x.double().numpy(), *[q.detach().double().numpy() for q in w],
                               dtype=np.float64)
        scale = np.abs(ref64).max()
        assert np.abs(got32 - ref64).max() <= 1e-5 * scale and np.abs(ref32 - ref64).max() <= 1e-5 * scale
        assert np.abs(got64 - ref64).max() <= 1e-12 * scale
    # leading dims are kept
    assert oracle.mlp_fwd(np.zeros((2, 3, 8), np.float32), np.zeros((128, 8), np.float32),
                          np.zeros(128, np.float32), np.zeros((64, 128), np.float32),
                          np.ones(64, np.float32)).shape == (2, 3, 64)
