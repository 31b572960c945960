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
    main, l1, ssim = losses.main_loss(r, g, vm, lam)
    np.testing.assert_allclose([main.item(), l1.item(), ssim.item()], ref, rtol=2e-6, atol=1e-7)
    assert not l1.requires_grad and not ssim.requires_grad
    (main * 1.7).backward()
    v_ref = oracle.image_loss_bwd(rgb, gt, valid, lam, 1.7)
    np.testing.assert_array_equal(r.grad.cpu().numpy(), v_ref)          # bit for bit
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
