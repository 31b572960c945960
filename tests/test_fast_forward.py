"""The batched pair forward (round 4, gg_blend_fwd_pair_fast / csrc/blend2.hip blend2_fwd_batch_kernel): the product's
DEFAULT forward walk of the fused operator (feature 32 | rgb + depth + normal).

Contract (include/gg_raster.h): final_Ts, final_idx and with them every alpha / stop decision are the exact-order
kernel's and the oracle's, BIT FOR BIT; the images agree to fp32 rounding — here held to

        |image - oracle| <= 1e-6 (1 + |oracle|)          (BASELINE asks for 1e-5 max-abs)

on colours of ordinary range, and to the channel-wise bound 2^-20 x (largest |colour| of the channel) — sixteen fp32
ulps of it: two summation orders of some tens of fp32 terms — for colour arrays spread over six decades.  The reference side of these calls is gsplat's rasterize_forward / nd_rasterize_forward (call
sites nerfstudio/models/gaussian_splatting.py:735-784); the oracle is oracle/gg_oracle.c (parity UNPINNED against the
real gsplat: PARITY.md).  Every other test of the suite runs with ops.EXACT_FORWARD = True (tests/conftest.py), i.e. on
the exact-order kernel whose images are the oracle's bit for bit."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

DEV = "cuda:0"
TOL = 1e-6


def _t(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(DEV)


def _np(t):
    return t.detach().cpu().numpy()


def _close(img, ref, what, tol=TOL):
    img, ref = np.asarray(img, np.float64), np.asarray(ref, np.float64)
    excess = np.abs(img - ref) - tol * (1.0 + np.abs(ref))
    assert excess.max() <= 0.0, f"{what}: max |diff| {np.abs(img - ref).max():.3e}, worst excess {excess.max():.3e}"
    return float(np.abs(img - ref).max())


class Pair:
    """feature (C >= 32) | second array (c2 <= 8) through the C ABI, exact and fast"""

    def __init__(self, oracle, n, h, w, c, c2, seed, colors=None):
        from gaussiangrasper_amd import _lib, ops as P
        from test_gpu_parity import _blend_inputs
        self.lib, self.P, self._lib = _lib.load(), P, _lib
        self.n, self.h, self.w, self.c, self.c2 = n, h, w, c, c2
        xys, depths, radii, conics, nth, col, opac, bg = _blend_inputs(oracle, n, h, w, c + c2, seed=seed)
        if colors is not None:
            col = colors(col)
        self.np_in = (xys, depths, radii, conics, nth, col, opac, bg)
        self.xys, self.conics, self.opac = _t(xys), _t(conics), _t(opac)
        self.col, self.col2 = _t(col[:, :c]), _t(col[:, c:])
        self.bg, self.bg2 = _t(bg[:c]), _t(bg[c:])
        P.clear_bin_cache()
        b = P.bin_and_sort_gaussians(self.xys, _t(depths), _t(radii), _t(nth), h, w)
        self.ids, self.tile_bins = b.gaussian_ids_sorted, b.tile_bins
        self.ws = torch.empty(self.lib.gg_blend_workspace(n), dtype=torch.uint8, device=DEV)

    def run(self, fast):
        lib, p = self.lib, self.P._ptr
        h, w = self.h, self.w
        img = torch.full((h, w, self.c), -3.0, device=DEV)
        img2 = torch.full((h, w, self.c2), -3.0, device=DEV)
        fT = torch.full((h, w), -3.0, device=DEV)
        fi = torch.full((h, w), -3, dtype=torch.int32, device=DEV)
        fn = lib.gg_blend_fwd_pair_fast if fast else lib.gg_blend_fwd_pair
        self._lib.check(fn(self.c, self.c2, self.n, h, w, p(self.ids), p(self.tile_bins), p(self.xys), p(self.conics),
                           p(self.col), p(self.col2), p(self.opac), p(self.bg), p(self.bg2), p(img), p(img2), p(fT), p(fi),
                           p(self.ws), self.ws.numel(), self.P._stream(self.xys.device)), "gg_blend_fwd_pair*")
        torch.cuda.synchronize()
        return img, img2, fT, fi

    def oracle_images(self, oracle):
        xys, depths, radii, conics, nth, col, opac, bg = self.np_in
        out = []
        for a, b in ((col[:, :self.c], bg[:self.c]), (col[:, self.c:], bg[self.c:])):
            img, saved = oracle.rasterize_fwd(xys, depths, radii, conics, nth, np.ascontiguousarray(a), opac, self.h, self.w,
                                              np.ascontiguousarray(b))
            out.append((img, saved))
        return out


@pytest.mark.parametrize("n,h,w,c,c2", [(1, 16, 16, 32, 7), (7, 45, 70, 32, 3), (2000, 48, 64, 32, 7), (50000, 300, 400, 32, 7),
                                        (20000, 150, 200, 32, 1), (4000, 83, 101, 32, 8), (4000, 90, 100, 64, 7),
                                        (3000, 64, 80, 36, 5), (300000, 600, 800, 32, 7)])
def test_fast_pair_forward_decisions_bitexact_images_to_fp32_rounding(oracle, n, h, w, c, c2):
    pc = Pair(oracle, n, h, w, c, c2, seed=23)
    e_img, e_img2, e_T, e_i = pc.run(False)
    f_img, f_img2, f_T, f_i = pc.run(True)
    assert torch.equal(e_T, f_T) and torch.equal(e_i, f_i), "final_T / final_idx must keep their bits"
    (o1, s1), (o2, _) = pc.oracle_images(oracle)
    assert np.array_equal(_np(f_T).view(np.uint32), s1["final_Ts"].view(np.uint32))
    assert np.array_equal(_np(f_i), s1["final_idx"])
    _close(_np(f_img), o1, "first array vs oracle")
    _close(_np(f_img2), o2, "second array vs oracle")
    _close(_np(f_img), _np(e_img), "first array vs exact kernel")
    if c > 32:      # channels beyond the first 32 are walked by the exact block kernels: bits
        assert torch.equal(f_img[..., 32:], e_img[..., 32:])


def test_fast_pair_forward_against_a_float64_sum(oracle):
    """the batched kernel's images are as close to the fp64 oracle as the exact-order fp32 kernel's are"""
    pc = Pair(oracle, 30000, 200, 260, 32, 7, seed=29)
    e_img, e_img2, _, _ = pc.run(False)
    f_img, f_img2, _, _ = pc.run(True)
    xys, depths, radii, conics, nth, col, opac, bg = pc.np_in
    f64 = lambda a: np.asarray(a, np.float64)
    errs = {}
    for name, a, b, ei, fi in (("first", col[:, :32], bg[:32], e_img, f_img), ("second", col[:, 32:], bg[32:], e_img2, f_img2)):
        ref, _ = oracle.rasterize_fwd(f64(xys), f64(depths), radii, f64(conics), nth, f64(np.ascontiguousarray(a)), f64(opac),
                                      pc.h, pc.w, f64(np.ascontiguousarray(b)), dtype=np.float64)
        # (the fp64 oracle makes its own alpha / stop decisions: compare where they agree, i.e. nearly everywhere)
        de, df = np.abs(_np(ei) - ref), np.abs(_np(fi) - ref)
        same = de.max(axis=-1) < 1e-4
        assert same.mean() > 0.999
        errs[name] = (float(de[same].max()), float(df[same].max()))
        assert df[same].max() <= max(2.0 * de[same].max(), 5e-7), (name, errs[name])
    print("max |error| against the fp64 oracle (exact-order kernel, batched kernel):", errs)


@pytest.mark.parametrize("spread", ["channels", "rows", "tiny", "huge"])
def test_fast_pair_forward_over_a_wide_dynamic_range(oracle, spread):
    """colour arrays spread over six decades: per channel the error stays below 2^-20 of the channel's largest |colour|
    (the colour scale is per channel: a channel of 1e-3 next to one of 1e3 keeps its own relative accuracy)"""
    def colors(col):
        rng = np.random.default_rng(3)
        col = col.copy()
        if spread == "channels":
            col *= (10.0 ** rng.uniform(-3, 3, col.shape[1])).astype(np.float32)[None, :]
        elif spread == "rows":
            col *= (10.0 ** rng.uniform(-3, 3, col.shape[0])).astype(np.float32)[:, None]
        elif spread == "tiny":
            col *= np.float32(1e-20)
        else:
            col *= np.float32(1e15)
        return col
    pc = Pair(oracle, 20000, 150, 200, 32, 7, seed=31, colors=colors)
    _, _, e_T, e_i = pc.run(False)
    f_img, f_img2, f_T, f_i = pc.run(True)
    assert torch.equal(e_T, f_T) and torch.equal(e_i, f_i)
    (o1, _), (o2, _) = pc.oracle_images(oracle)
    col, bg = pc.np_in[5], pc.np_in[7]
    for img, ref, cc, bb in ((_np(f_img), o1, col[:, :32], bg[:32]), (_np(f_img2), o2, col[:, 32:], bg[32:])):
        cmax = np.maximum(np.abs(cc).max(axis=0), np.abs(bb)).astype(np.float64)       # per channel
        err = np.abs(img.astype(np.float64) - ref).reshape(-1, img.shape[-1]).max(axis=0)
        assert (err <= 2.0 ** -20 * cmax + 1e-37).all(), (spread, (err / cmax).max())


def test_fast_forward_is_the_operators_default_and_backward_is_unchanged(oracle):
    """ops.rasterize_segments: fast by default, exact with ops.set_exact_forward(True); images to fp32 rounding, and — the
    backward reads final_T / final_idx only — gradients equal to atomics noise"""
    from gaussiangrasper_amd import ops as P
    from test_gpu_parity import _blend_inputs
    n, h, w = 8000, 90, 120
    xys, depths, radii, conics, nth, colors, opac, bg = _blend_inputs(oracle, n, h, w, 39, 47)
    outs = {}
    prev = P.EXACT_FORWARD
    try:
        for exact in (True, False):
            P.set_exact_forward(exact)
            xt, ct, ot = (_t(a).requires_grad_(True) for a in (xys, conics, opac))
            c1, c2 = _t(colors[:, :32]).requires_grad_(True), _t(colors[:, 32:]).requires_grad_(True)
            P.clear_bin_cache()
            imgs = P.rasterize_segments(xt, _t(depths), _t(radii), ct, _t(nth), ot, h, w,
                                        [(c1, _t(bg[:32])), (c2, _t(bg[32:]))])
            g = torch.Generator(device="cpu").manual_seed(3)
            cots = [torch.randn(i.shape, generator=g).to(DEV) for i in imgs]
            torch.autograd.backward(imgs, cots)
            outs[exact] = ([i.detach().clone() for i in imgs], [t.grad.clone() for t in (xt, ct, ot, c1, c2)])
    finally:
        P.set_exact_forward(prev)
    assert prev is True, "tests/conftest.py runs the suite on the exact-order kernel"
    differ = False
    for a, b in zip(outs[False][0], outs[True][0]):
        _close(_np(a), _np(b), "operator image, fast vs exact")
        differ = differ or not torch.equal(a, b)
    assert differ, "the default forward should be the batched kernel (images differ from the exact one in the last bits)"
    for a, b in zip(outs[False][1], outs[True][1]):
        scale = float(b.abs().max())
        assert float((a - b).abs().max()) <= 2e-6 * scale + 1e-30


def test_fast_forward_through_the_plugin_class_at_full_size(oracle):
    """BASELINE configs 3 / 4 geometry at 1600x1200 through the class train.sh loads, default (fast) forward: the four
    images against the exact-order run of the same class (whose images are the oracle's bits:
    test_gpu_parity.py::test_full_size_vs_oracle, test_plugin.py)"""
    from gaussiangrasper_amd import ops as P
    from gaussiangrasper_amd.camera import ring_cameras
    from gaussiangrasper_amd.plugin import make_fused_model_class
    from gaussiangrasper_amd.scene import make_scene
    from gaussiangrasper_amd.stub import StubCameras, StubGaussianSplattingModel
    h, w = 1200, 1600
    sc = make_scene(300_000, config_index=2).to(DEV)
    m = make_fused_model_class(StubGaussianSplattingModel)(sc).to(DEV).eval()
    cam = lambda: StubCameras.from_view(ring_cameras(4, h, w, device=DEV)[1], device=DEV)
    prev = P.EXACT_FORWARD
    try:
        P.set_exact_forward(True)
        with torch.no_grad():
            ex = {k: v.clone() for k, v in m.get_outputs(cam()).items() if k in ("rgb", "feature", "depth", "normal")}
        P.set_exact_forward(False)
        with torch.no_grad():
            fa = {k: v.clone() for k, v in m.get_outputs(cam()).items() if k in ("rgb", "feature", "depth", "normal")}
    finally:
        P.set_exact_forward(prev)
    worst = {k: _close(_np(fa[k]), _np(ex[k]), k) for k in ex}
    print("plugin class at 1600x1200, fast vs exact forward, max |diff|:", worst)

