"""Image-space main loss of get_loss_dict (reference gaussian_splatting.py:882-885, :931; SSIM :284):
(1 - lambda) L1 + lambda (1 - SSIM).  pytorch-msssim (requirements.txt:199, ==1.0.0) is not in the tree and
not installed, so SSIM is PARITY UNPINNED against the package; what is pinned: the oracle restatement against torch
autograd of the package's published algorithm written with F.conv2d (below), in fp64 and fp32, with the reference's
masking and reduction around it; the HIP kernels against the oracle (maps and gradient image bit for bit)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F


def _ssim_published(X, Y):
    """pytorch_msssim 1.0.0 `ssim(X, Y, data_range=1, size_average=True)` for (1, 3, H, W): 11-tap Gaussian window,
    sigma 1.5, valid separable filtering (H first, then W), K = (0.01, 0.03)."""
    size, sigma = 11, 1.5
    coords = torch.arange(size, dtype=X.dtype) - size // 2
    g = torch.exp(-(coords ** 2) / (2 * sigma ** 2))
    g = g / g.sum()
    win = g.reshape(1, 1, 1, -1).repeat(3, 1, 1, 1)

    def gf(t):
        return F.conv2d(F.conv2d(t, win.transpose(2, -1), groups=3), win, groups=3)
    C1, C2 = 0.01 ** 2, 0.03 ** 2
    mu1, mu2 = gf(X), gf(Y)
    s1, s2, s12 = gf(X * X) - mu1 ** 2, gf(Y * Y) - mu2 ** 2, gf(X * Y) - mu1 * mu2
    cs = (2 * s12 + C2) / (s1 + s2 + C2)
    return (((2 * mu1 * mu2 + C1) / (mu1 ** 2 + mu2 ** 2 + C1)) * cs).flatten(2).mean(-1).mean()


def _reference_main_loss(rgb, gt, valid, lam):
    """the lines of get_loss_dict, on clones (the reference zeroes its arguments in place)"""
    gt, out = gt.clone(), rgb
    if valid is None:
        valid = torch.ones(rgb.shape[:2], dtype=torch.bool)
    Ll1 = torch.abs(gt[valid, :] - out[valid, :]).mean()
    gt[~valid, :] = 0.0
    out = out.clone()
    out[~valid, :] = 0.0
    ssim = _ssim_published(gt.permute(2, 0, 1)[None, ...], out.permute(2, 0, 1)[None, ...])
    return (1 - lam) * Ll1 + lam * (1 - ssim), Ll1, ssim


def _images(h, w, seed, dtype=np.float32, masked=True):
    rng = np.random.default_rng(seed)
    gt = rng.uniform(0, 1, (h, w, 3)).astype(dtype)
    yy, xx = np.mgrid[0:h, 0:w]
    gt = (0.6 * gt + 0.4 * (0.5 + 0.5 * np.sin(yy / 5.0 + xx / 7.0))[..., None]).astype(dtype)   # some structure
    rgb = np.clip(gt + rng.normal(0, 0.08, (h, w, 3)), 0, 1).astype(dtype)
    valid = (rng.uniform(size=(h, w)) > 0.15) if masked else None
    return rgb, gt, valid


@pytest.mark.parametrize("h,w,masked", [(11, 11, False), (37, 45, True), (64, 80, False), (50, 33, True)])
@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_oracle_main_loss_matches_torch_autograd_of_the_published_algorithm(oracle, h, w, masked, dtype):
    rgb, gt, valid = _images(h, w, 3, dtype, masked)
    lam = 0.2
    out = oracle.image_loss_fwd(rgb, gt, valid, lam, dtype=dtype)
    r = torch.from_numpy(rgb.copy()).requires_grad_(True)
    main, l1, ssim = _reference_main_loss(r, torch.from_numpy(gt.copy()), None if valid is None else torch.from_numpy(valid),
                                          lam)
    tol = 1e-12 if dtype == np.float64 else 2e-6
    np.testing.assert_allclose(out, [main.item(), l1.item(), ssim.item()], rtol=tol, atol=tol)
    (main * 1.7).backward()
    v = oracle.image_loss_bwd(rgb, gt, valid, lam, 1.7, dtype=dtype)
    scale = np.abs(r.grad.numpy()).max()
    assert np.abs(v - r.grad.numpy()).max() <= (1e-12 if dtype == np.float64 else 2e-5) * scale   # fp32: sigma = E[y^2] - mu^2 cancels
    if valid is not None:
        assert (v[~valid] == 0).all()


@pytest.mark.gpu
@pytest.mark.parametrize("h,w,masked", [(11, 11, False), (16, 27, True), (96, 128, True), (300, 400, False),
                                        (1200, 1600, True)])
def test_hip_main_loss_vs_oracle(oracle, h, w, masked):
    from gaussiangrasper_amd import losses
    dev = torch.device("cuda:0")
    rgb, gt, valid = _images(h, w, 9, np.float32, masked)
    lam = 0.2
    ref = oracle.image_loss_fwd(rgb, gt, valid, lam)
    r = torch.from_numpy(rgb).to(dev).requires_grad_(True)
    g = torch.from_numpy(gt).to(dev)
    vm = None if valid is None else torch.from_numpy(valid).to(dev)
    g_before = g.clone()
    if masked:      # the plugin route's rgb: channels 0..2 of an (H, W, 7) image, read in place
        wide = torch.zeros(h, w, 7, device=dev)
        wide[..., :3] = r.detach()
        wide.requires_grad_(True)
        m2 = losses.main_loss(wide[..., :3], g, vm, lam)[0]
        (m2 * 1.7).backward()
        sliced_grad = wide.grad[..., :3].clone()
        assert not wide.grad[..., 3:].any()
    main, l1, ssim = losses.main_loss(r, g, vm, lam)
    np.testing.assert_allclose([main.item(), l1.item(), ssim.item()], ref, rtol=2e-6, atol=1e-7)
    assert not l1.requires_grad and not ssim.requires_grad
    (main * 1.7).backward()
    v_ref = oracle.image_loss_bwd(rgb, gt, valid, lam, 1.7)
    np.testing.assert_array_equal(r.grad.cpu().numpy(), v_ref)          # bit for bit
    if masked:
        assert m2.item() == main.item() and torch.equal(sliced_grad, r.grad)
    assert torch.equal(g, g_before) and torch.equal(r.detach().cpu(), torch.from_numpy(rgb))   # no side effects
    # and the caller's torch ops on the same device agree
    r2 = torch.from_numpy(rgb).to(dev).requires_grad_(True)
    m2, l2, s2 = _reference_main_loss_dev(r2, g, vm, lam)
    np.testing.assert_allclose([main.item(), l1.item(), ssim.item()], [m2.item(), l2.item(), s2.item()], rtol=5e-6,
                               atol=1e-7)
    (m2 * 1.7).backward()
    scale = r2.grad.abs().max().item()
    assert (r.grad - r2.grad).abs().max().item() <= 2e-5 * scale


def _reference_main_loss_dev(rgb, gt, valid, lam):
    gt = gt.clone()
    if valid is None:
        valid = torch.ones(rgb.shape[:2], dtype=torch.bool, device=rgb.device)
    Ll1 = torch.abs(gt[valid, :] - rgb[valid, :]).mean()
    gt[~valid, :] = 0.0
    out = rgb.clone()
    out[~valid, :] = 0.0
    X, Y = gt.permute(2, 0, 1)[None, ...], out.permute(2, 0, 1)[None, ...]
    size, sigma = 11, 1.5
    coords = torch.arange(size, dtype=X.dtype, device=X.device) - size // 2
    gk = torch.exp(-(coords ** 2) / (2 * sigma ** 2))
    gk = gk / gk.sum()
    win = gk.reshape(1, 1, 1, -1).repeat(3, 1, 1, 1)
    gf = lambda t: F.conv2d(F.conv2d(t, win.transpose(2, -1), groups=3), win, groups=3)
    C1, C2 = 0.01 ** 2, 0.03 ** 2
    mu1, mu2 = gf(X), gf(Y)
    s1, s2, s12 = gf(X * X) - mu1 ** 2, gf(Y * Y) - mu2 ** 2, gf(X * Y) - mu1 * mu2
    ssim = (((2 * mu1 * mu2 + C1) / (mu1 ** 2 + mu2 ** 2 + C1)) * ((2 * s12 + C2) / (s1 + s2 + C2))).flatten(2).mean(
        -1).mean()
    return (1 - lam) * Ll1 + lam * (1 - ssim), Ll1, ssim


@pytest.mark.gpu
def test_main_loss_error_behaviour():
    from gaussiangrasper_amd import losses
    dev = torch.device("cuda:0")
    with pytest.raises(ValueError):
        losses.main_loss(torch.zeros(8, 8, 3, device=dev), torch.zeros(8, 8, 3, device=dev))
    with pytest.raises(ValueError):
        losses.main_loss(torch.zeros(20, 20, 3, device=dev), torch.zeros(20, 21, 3, device=dev))
    with pytest.raises(ValueError):
        losses.main_loss(torch.zeros(20, 20, 3, device=dev), torch.zeros(20, 20, 3, device=dev),
                         torch.ones(20, 19, dtype=torch.bool, device=dev))


# ---- depth and normal losses (reference :879-880) -------------------------------------------------------------
def _geom_inputs(h, w, seed, dtype=np.float32):
    rng = np.random.default_rng(seed)
    depth = rng.uniform(0.5, 5, (h, w, 1)).astype(dtype)
    gt_depth = rng.uniform(0.5, 5, (h, w)).astype(dtype)
    normal = rng.normal(size=(h, w, 3)).astype(dtype)
    gt_normal = rng.normal(size=(3, h, w)).astype(dtype)
    gt_normal /= np.linalg.norm(gt_normal, axis=0, keepdims=True)
    mask = rng.uniform(size=(h, w)) > 0.3
    normal[0, 0] = 0.0            # a zero vector: F.normalize's eps branch
    mask[0, 0] = True
    return depth, gt_depth, normal, gt_normal, mask


def _reference_geom_losses(depth, gt_depth, normal, gt_normal, mask):
    """the two lines of get_loss_dict; depth (H, W, 1), normal (H, W, 3) model outputs; gt_normal (3, H, W)"""
    def cosine_similarity_loss(e1, e2):
        return 1 - (F.normalize(e1, dim=0) * F.normalize(e2, dim=0)).sum(dim=0).mean()
    n, d = normal.permute(2, 0, 1), depth.permute(2, 0, 1)
    m = mask[None]
    normal_loss = 0.5 * F.mse_loss(n[:, m[0]], gt_normal[:, m[0]], reduction="mean") + \
        0.5 * cosine_similarity_loss(n[:, m[0]], gt_normal[:, m[0]])
    depth_loss = F.l1_loss(d[m], gt_depth[None][m], reduction="mean")
    return depth_loss, normal_loss


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_oracle_depth_normal_losses_match_torch_autograd_of_the_reference_lines(oracle, dtype):
    h, w = 23, 31
    depth, gt_depth, normal, gt_normal, mask = _geom_inputs(h, w, 1, dtype)
    out = oracle.geom_loss_fwd(depth, gt_depth, normal, gt_normal, mask, dtype=dtype)
    d, n = torch.from_numpy(depth).requires_grad_(True), torch.from_numpy(normal).requires_grad_(True)
    dl, nl = _reference_geom_losses(d, torch.from_numpy(gt_depth), n, torch.from_numpy(gt_normal), torch.from_numpy(mask))
    tol = 1e-12 if dtype == np.float64 else 2e-6
    np.testing.assert_allclose(out[:2], [dl.item(), nl.item()], rtol=tol, atol=tol)
    assert out[2] == mask.sum()
    (1.3 * dl + 0.7 * nl).backward()
    vd, vn = oracle.geom_loss_bwd(depth, gt_depth, normal, gt_normal, mask, 1.3, 0.7, dtype=dtype)
    atol = 1e-15 if dtype == np.float64 else 1e-9
    np.testing.assert_allclose(vd.reshape(h, w, 1), d.grad.numpy(), rtol=tol * 10, atol=atol)
    np.testing.assert_allclose(vn.reshape(h, w, 3), n.grad.numpy(), rtol=tol * 10, atol=atol)


@pytest.mark.gpu
@pytest.mark.parametrize("h,w,sliced", [(23, 31, False), (96, 128, True), (1200, 1600, True)])
def test_hip_depth_normal_losses_vs_oracle(oracle, h, w, sliced):
    from gaussiangrasper_amd import losses
    dev = torch.device("cuda:0")
    depth, gt_depth, normal, gt_normal, mask = _geom_inputs(h, w, 4)
    ref = oracle.geom_loss_fwd(depth, gt_depth, normal, gt_normal, mask)
    if sliced:      # the plugin route's outputs: channel slices of one (H, W, 7) image
        tail = torch.zeros(h, w, 7, device=dev)
        tail[..., 3:4] = torch.from_numpy(depth).to(dev)
        tail[..., 4:7] = torch.from_numpy(normal).to(dev)
        tail.requires_grad_(True)
        d, n = tail[..., 3:4], tail[..., 4:7]
    else:
        d = torch.from_numpy(depth).to(dev).requires_grad_(True)
        n = torch.from_numpy(normal).to(dev).requires_grad_(True)
    dl, nl = losses.depth_normal_loss(d, torch.from_numpy(gt_depth).to(dev), n, torch.from_numpy(gt_normal).to(dev),
                                      torch.from_numpy(mask).to(dev))
    np.testing.assert_allclose([dl.item(), nl.item()], ref[:2], rtol=2e-6, atol=1e-7)
    (1.3 * dl + 0.7 * nl).backward()
    vd, vn = oracle.geom_loss_bwd(depth, gt_depth, normal, gt_normal, mask, 1.3, 0.7)
    if sliced:
        g = tail.grad.cpu().numpy()
        got_d, got_n = g[..., 3], g[..., 4:7]
        assert (g[..., :3] == 0).all()
    else:
        got_d, got_n = d.grad.cpu().numpy()[..., 0], n.grad.cpu().numpy()
    np.testing.assert_array_equal(got_d.reshape(-1), vd)                  # bit for bit
    np.testing.assert_array_equal(got_n.reshape(-1, 3), vn)
