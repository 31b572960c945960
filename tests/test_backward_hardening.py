"""The pair entries through the C ABI against the oracle, and the hardening of the backward entry points against a
caller's bad `final_idx` (the fault in gpurun_out/stamps_skip1.log came from an uninitialised
final_idx image).  (Round 3's quad lists — the forward's cull survivors persisted for the backward — measured no gain
and left the library in round 4: profiles/r03_quad_lists_experiment.patch.)

The reference side of these calls is gsplat's rasterize_backward (call sites: reference
nerfstudio/models/gaussian_splatting.py:735-784 under autograd); the oracle is oracle/gg_oracle.c."""
import ctypes as C

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

DEV = "cuda:0"


@pytest.fixture(scope="module")
def oracle():
    from oracle import oracle as O
    O.build()
    return O


def _t(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(DEV)


def _np(t):
    return t.detach().cpu().numpy()


def _inputs(oracle, n, h, w, ch, seed):
    from test_gpu_parity import _blend_inputs
    return _blend_inputs(oracle, n, h, w, ch, seed=seed)


class PairCall:
    """the two colour arrays of a fused view (32 + c2 channels) through the C ABI"""

    def __init__(self, oracle, n, h, w, c2, seed):
        from gaussiangrasper_amd import _lib, ops as P
        self.lib, self.P, self._lib = _lib.load(), P, _lib
        self.n, self.h, self.w, self.c2 = n, h, w, c2
        xys, depths, radii, conics, nth, colors, opac, bg = _inputs(oracle, n, h, w, 32 + c2, seed)
        self.np_in = (xys, depths, radii, conics, nth, colors, opac, bg)
        self.xys, self.conics, self.opac = _t(xys), _t(conics), _t(opac)
        self.col, self.col2 = _t(colors[:, :32]), _t(colors[:, 32:])
        self.bg, self.bg2 = _t(bg[:32]), _t(bg[32:])
        P.clear_bin_cache()
        bins = P.bin_and_sort_gaussians(self.xys, _t(depths), _t(radii), _t(nth), h, w)
        self.ids, self.tile_bins, self.I = bins.gaussian_ids_sorted, bins.tile_bins, bins.num_intersects
        self.ws = torch.empty(self.lib.gg_blend_workspace(n), dtype=torch.uint8, device=DEV)
        self.stream = P._stream(self.xys.device)

    def forward(self):
        lib, p = self.lib, self.P._ptr
        h, w = self.h, self.w
        self.img = torch.empty(h, w, 32, device=DEV)
        self.img2 = torch.empty(h, w, self.c2, device=DEV)
        self.fT = torch.empty(h, w, device=DEV)
        self.fi = torch.empty(h, w, dtype=torch.int32, device=DEV)
        head = (32, self.c2, self.n, h, w, p(self.ids), p(self.tile_bins), p(self.xys), p(self.conics), p(self.col),
                p(self.col2), p(self.opac), p(self.bg), p(self.bg2), p(self.img), p(self.img2), p(self.fT),
                p(self.fi), p(self.ws), self.ws.numel())
        self._lib.check(lib.gg_blend_fwd_pair(*head, self.stream), "gg_blend_fwd_pair")
        return self.img, self.img2

    def backward(self, v1, v2, final_idx=None, expect_ok=True):
        lib, p = self.lib, self.P._ptr
        n = self.n
        rec = torch.empty(n, 6 + self.c2, device=DEV)
        vcol = torch.empty(n, 32, device=DEV)
        parts = (C.c_void_p * 1)(p(v2))
        chs = (C.c_int * 1)(self.c2)
        fi = self.fi if final_idx is None else final_idx
        head = (32, self.c2, n, self.h, self.w, p(self.ids), p(self.tile_bins), p(self.xys), p(self.conics),
                p(self.col), p(self.col2), p(self.opac), p(self.bg), p(self.bg2), p(self.fT), p(fi), p(v1), parts,
                chs, 1, p(rec[:, 0:2]), p(rec[:, 2:5]), p(vcol), p(rec[:, 6:]), p(rec[:, 5:6]), 6 + self.c2, 0,
                6 + self.c2, p(self.ws), self.ws.numel(), 1)
        st = lib.gg_blend_bwd_pair(*head, self.stream)
        if expect_ok:
            self._lib.check(st, "gg_blend_bwd_pair")
        torch.cuda.synchronize()
        return st, rec, vcol

@pytest.mark.parametrize("n,h,w,c2", [(6000, 77, 101, 7), (20000, 150, 200, 1), (40000, 300, 400, 7)])
def test_pair_backward_through_the_c_abi_matches_the_oracle(oracle, n, h, w, c2):
    """gg_blend_bwd_pair (interleaved record of 6 + c2 floats: the separate-atomics build) against the oracle's two
    separate backward calls, with the tolerance of test_blend_bwd"""
    from test_gpu_parity import assert_close
    pc = PairCall(oracle, n, h, w, c2, seed=43)
    xys, depths, radii, conics, nth, colors, opac, bg = pc.np_in
    rng = np.random.default_rng(9)
    ref = None
    vs = []
    for col, b in ((colors[:, :32], bg[:32]), (colors[:, 32:], bg[32:])):
        out, saved = oracle.rasterize_fwd(xys, depths, radii, conics, nth, np.ascontiguousarray(col), opac, h, w,
                                          np.ascontiguousarray(b))
        v = rng.standard_normal(out.shape).astype(np.float32)
        vs.append(v)
        bb = saved["bins"]
        g = oracle.blend_bwd(bb["gaussian_ids_sorted"], bb["tile_bins"], xys, conics, np.ascontiguousarray(col), opac,
                             h, w, np.ascontiguousarray(b), saved["final_Ts"], saved["final_idx"], v)
        ref = [g[0].astype(np.float64), g[1].astype(np.float64), [g[2]], g[3].astype(np.float64)] if ref is None \
            else [ref[0] + g[0], ref[1] + g[1], ref[2] + [g[2]], ref[3] + g[3]]
    pc.forward()
    v1, v2 = _t(vs[0]), _t(vs[1])
    _, rec_w, col_w = pc.backward(v1, v2)
    for name, rec, col in (("walk", rec_w, col_w),):
        r = _np(rec)
        assert_close(r[:, 0:2], ref[0], f"{name}.v_xy", rtol=5e-5, atol_frac=1e-6)
        assert_close(r[:, 2:5], ref[1], f"{name}.v_conic", rtol=5e-5, atol_frac=1e-6)
        assert_close(r[:, 5:6], ref[3].reshape(-1, 1), f"{name}.v_opacity", rtol=5e-5, atol_frac=1e-6)
        assert_close(r[:, 6:], ref[2][1], f"{name}.v_colors2", rtol=5e-5, atol_frac=1e-6)
        assert_close(_np(col), ref[2][0], f"{name}.v_colors", rtol=5e-5, atol_frac=1e-6)


@pytest.mark.parametrize("ch", [3, 8, 32])
@pytest.mark.parametrize("bad", [2 ** 31 - 1, -1, -(2 ** 31)])
def test_blend_bwd_survives_a_garbage_final_idx(oracle, ch, bad):
    """gg_blend_bwd is a public entry taking the caller's final_idx: an image of INT_MAX / -1 / INT_MIN must give
    status 0 and no GPU memory fault (the kernels hold final_idx to the tile's list range).  Below every list nothing
    is walked (zero gradients); above, the walk covers entries the forward never blended, so T = T_final / prod(1 -
    alpha) is meaningless and may overflow: the values are then garbage-in-garbage-out, what is checked is that the
    call completes and the device still answers a correct call afterwards."""
    from gaussiangrasper_amd import _lib, ops as P
    lib = _lib.load()
    n, h, w = 3000, 64, 80
    xys, depths, radii, conics, nth, colors, opac, bg = _inputs(oracle, n, h, w, ch, 5)
    out, saved = oracle.rasterize_fwd(xys, depths, radii, conics, nth, colors, opac, h, w, bg)
    b = saved["bins"]
    ids, bins_t = _t(b["gaussian_ids_sorted"].astype(np.int32)), _t(b["tile_bins"].astype(np.int32))
    xt, ct, colt, ot, bgt = _t(xys), _t(conics), _t(colors), _t(opac), _t(bg)
    ft = _t(saved["final_Ts"])
    fi = torch.full((h, w), bad, dtype=torch.int32, device=DEV)
    vt = torch.randn(h, w, ch, device=DEV)
    ws = torch.empty(lib.gg_blend_workspace(n), dtype=torch.uint8, device=DEV)
    vx, vc, vo_ = (torch.empty(n, k, device=DEV) for k in (2, 3, 1))
    vcol = torch.empty(n, ch, device=DEV)
    p = P._ptr
    st = lib.gg_blend_bwd(ch, n, h, w, p(ids), p(bins_t), p(xt), p(ct), p(colt), p(ot), p(bgt), p(ft), p(fi), p(vt),
                          p(vx), p(vc), p(vcol), p(vo_), 0, 0, p(ws), ws.numel(), 0, P._stream(xt.device))
    torch.cuda.synchronize()
    assert st == 0
    if bad < 0:   # final_idx below every list: nothing is walked
        for g in (vx, vc, vcol, vo_):
            assert bool(torch.isfinite(g).all())
        assert float(vx.abs().max()) == 0.0 and float(vcol.abs().max()) == 0.0
    # the device is alive and the same entry still computes the right thing
    fi_ok = _t(saved["final_idx"].astype(np.int32))
    st = lib.gg_blend_bwd(ch, n, h, w, p(ids), p(bins_t), p(xt), p(ct), p(colt), p(ot), p(bgt), p(ft), p(fi_ok), p(vt),
                          p(vx), p(vc), p(vcol), p(vo_), 0, 0, p(ws), ws.numel(), 0, P._stream(xt.device))
    torch.cuda.synchronize()
    assert st == 0 and all(bool(torch.isfinite(g).all()) for g in (vx, vc, vcol, vo_))


@pytest.mark.parametrize("bad", [2 ** 31 - 1, -1])
def test_pair_backward_survives_a_garbage_final_idx(oracle, bad):
    """the pair entry: a garbage final_idx image gives status 0 and no fault (zero gradients when it lies below every
    list), and the device answers a correct call afterwards"""
    pc = PairCall(oracle, 3000, 64, 80, 7, seed=6)
    pc.forward()
    v1, v2 = torch.randn(64, 80, 32, device=DEV), torch.randn(64, 80, 7, device=DEV)
    fi_bad = torch.full((64, 80), bad, dtype=torch.int32, device=DEV)
    st, rec, col = pc.backward(v1, v2, final_idx=fi_bad)
    assert st == 0
    if bad < 0:
        assert float(rec.abs().max()) == 0.0 and float(col.abs().max()) == 0.0
    st, rec, col = pc.backward(v1, v2)     # alive, and right again with the forward's own final_idx
    assert st == 0 and bool(torch.isfinite(rec).all()) and bool(torch.isfinite(col).all())
