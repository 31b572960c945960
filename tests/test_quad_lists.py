"""Quad lists (include/gg_raster.h, round 3): the forward's per-quadrant cull survivors persisted for the backward
walk — gg_blend_fwd_pair_lists / gg_blend_bwd_pair_lists through the C ABI — and the hardening of the backward
entry points against a caller's bad `final_idx` / stale list counts (the fault in
gpurun_out/stamps_skip1.log came from an uninitialised final_idx image).

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
    """the two colour arrays of a fused view (32 + c2 channels) through the C ABI, with or without quad lists"""

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

    def forward(self, lists: bool):
        lib, p = self.lib, self.P._ptr
        h, w = self.h, self.w
        self.img = torch.empty(h, w, 32, device=DEV)
        self.img2 = torch.empty(h, w, self.c2, device=DEV)
        self.fT = torch.empty(h, w, device=DEV)
        self.fi = torch.empty(h, w, dtype=torch.int32, device=DEV)
        head = (32, self.c2, self.n, h, w, p(self.ids), p(self.tile_bins), p(self.xys), p(self.conics), p(self.col),
                p(self.col2), p(self.opac), p(self.bg), p(self.bg2), p(self.img), p(self.img2), p(self.fT),
                p(self.fi), p(self.ws), self.ws.numel())
        if lists:
            self.ql = torch.empty(lib.gg_quad_lists_workspace(self.I, h, w), dtype=torch.uint8, device=DEV)
            self._lib.check(lib.gg_blend_fwd_pair_lists(*head, self.I, p(self.ql), self.ql.numel(), self.stream),
                            "gg_blend_fwd_pair_lists")
        else:
            self._lib.check(lib.gg_blend_fwd_pair(*head, self.stream), "gg_blend_fwd_pair")
        return self.img, self.img2

    def backward(self, v1, v2, lists: bool, final_idx=None, expect_ok=True):
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
        if lists:
            st = lib.gg_blend_bwd_pair_lists(*head, self.I, p(self.ql), self.ql.numel(), self.stream)
        else:
            st = lib.gg_blend_bwd_pair(*head, self.stream)
        if expect_ok:
            self._lib.check(st, "gg_blend_bwd_pair*")
        torch.cuda.synchronize()
        return st, rec, vcol

    def list_counts_and_records(self):
        ntiles = self.tile_bins.shape[0]
        cnt_bytes = (ntiles * 16 + 255) // 256 * 256
        cnt = self.ql[:ntiles * 16].view(torch.int32).reshape(ntiles, 4)
        recs = self.ql[cnt_bytes:].view(torch.float32).reshape(-1, 8)
        return _np(cnt), recs


@pytest.mark.parametrize("n,h,w,c2", [(6000, 77, 101, 7), (20000, 150, 200, 7), (300, 40, 40, 3)])
def test_quad_lists_hold_exactly_the_backwards_survivors(oracle, n, h, w, c2):
    """The lists the forward writes, read back: every quadrant's records are in ascending list position inside the
    tile's range, carry the list's Gaussian id and that Gaussian's packed geometry, and contain every entry some
    pixel of the quadrant blended (final_idx and the alpha test recomputed on the host from the oracle's forward).
    The images are bit-identical with and without the lists."""
    pc = PairCall(oracle, n, h, w, c2, seed=41)
    a0, b0 = (x.clone() for x in pc.forward(False))
    fi0 = pc.fi.clone()
    a1, b1 = pc.forward(True)
    assert torch.equal(a0, a1) and torch.equal(b0, b1) and torch.equal(fi0, pc.fi)
    cnt, recs = pc.list_counts_and_records()
    recs = _np(recs)
    ids, bins = _np(pc.ids), _np(pc.tile_bins)
    xys, conics, opac = pc.np_in[0], pc.np_in[3], pc.np_in[6].reshape(-1)
    fin = _np(pc.fi)
    tiles_x = (w + 15) // 16
    total = 0
    for tile in range(bins.shape[0]):
        s, e = int(bins[tile, 0]), int(bins[tile, 1])
        ty, tx = divmod(tile, tiles_x)
        for q in range(4):
            k = int(cnt[tile, q])
            assert 0 <= k <= e - s
            total += k
            y0, x0 = ty * 16 + (q >> 1) * 8, tx * 16 + (q & 1) * 8
            fq = fin[y0:y0 + 8, x0:x0 + 8]
            if k == 0:
                # nothing kept: no pixel of the quadrant may have blended anything
                assert fq.size == 0 or int(fq.max()) <= s, (tile, q)
                continue
            r = recs[4 * s + q * (e - s): 4 * s + q * (e - s) + k]
            pos = r[:, 3].view(np.int32)
            gid = r[:, 7].view(np.int32)
            assert np.all(np.diff(pos) > 0) and pos[0] >= s and pos[-1] < e, (tile, q)
            assert pos[-1] < int(fq.max()) + 64, (tile, q)     # trimmed at the quadrant's largest final_idx (to a chunk)
            assert np.array_equal(gid, ids[pos]), (tile, q)
            assert np.array_equal(r[:, 0:2], xys[gid]) and np.array_equal(r[:, 4:7], conics[gid]), (tile, q)
            assert np.array_equal(r[:, 2], opac[gid]), (tile, q)
            # completeness: every list entry below final_idx whose alpha reaches 1/255 at some pixel of the quadrant
            ys, xs = np.mgrid[y0:min(y0 + 8, h), x0:min(x0 + 8, w)]
            if ys.size == 0:
                continue
            cand = ids[s:int(fq.max())]
            dx = xys[cand, 0][:, None] - xs.reshape(1, -1).astype(np.float32)
            dy = xys[cand, 1][:, None] - ys.reshape(1, -1).astype(np.float32)
            sig = 0.5 * (conics[cand, 0][:, None] * dx * dx + conics[cand, 2][:, None] * dy * dy) \
                + conics[cand, 1][:, None] * dx * dy
            alpha = np.minimum(0.999, opac[cand][:, None] * np.exp(-sig.astype(np.float64)))
            live = (sig >= 0) & (alpha >= 1.0 / 255 * (1 + 1e-5)) & \
                   ((np.arange(s, int(fq.max()))[:, None]) < fq.reshape(1, -1)[:, :ys.size])
            need = np.arange(s, int(fq.max()))[live.any(axis=1)]
            assert np.all(np.isin(need, pos)), (tile, q)
    assert total > 0


@pytest.mark.parametrize("n,h,w,c2", [(6000, 77, 101, 7), (20000, 150, 200, 1), (40000, 300, 400, 7)])
def test_backward_from_quad_lists_matches_the_oracle_and_the_list_walk(oracle, n, h, w, c2):
    """gg_blend_bwd_pair_lists against the oracle's two separate backward calls (the tolerance of test_blend_bwd)
    and against gg_blend_bwd_pair on the same inputs (float-atomic order is all that differs)"""
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
    pc.forward(True)
    v1, v2 = _t(vs[0]), _t(vs[1])
    _, rec_l, col_l = pc.backward(v1, v2, True)
    _, rec_w, col_w = pc.backward(v1, v2, False)
    for name, rec, col in (("lists", rec_l, col_l), ("walk", rec_w, col_w)):
        r = _np(rec)
        assert_close(r[:, 0:2], ref[0], f"{name}.v_xy", rtol=5e-5, atol_frac=1e-6)
        assert_close(r[:, 2:5], ref[1], f"{name}.v_conic", rtol=5e-5, atol_frac=1e-6)
        assert_close(r[:, 5:6], ref[3].reshape(-1, 1), f"{name}.v_opacity", rtol=5e-5, atol_frac=1e-6)
        assert_close(r[:, 6:], ref[2][1], f"{name}.v_colors2", rtol=5e-5, atol_frac=1e-6)
        assert_close(_np(col), ref[2][0], f"{name}.v_colors", rtol=5e-5, atol_frac=1e-6)


def test_operator_uses_quad_lists_and_the_switch_turns_them_off(oracle):
    """ops.RasterizeSegments: with USE_QUAD_LISTS the backward runs gg_blend_bwd_pair_lists, without it the list walk;
    the two gradients agree to float-atomic noise and the images bit for bit"""
    from gaussiangrasper_amd import ops as P
    n, h, w = 8000, 90, 120
    xys, depths, radii, conics, nth, colors, opac, bg = _inputs(oracle, n, h, w, 39, 47)
    outs = {}
    for on in (True, False):
        prev, P.USE_QUAD_LISTS = P.USE_QUAD_LISTS, on
        try:
            xt, ct, ot = (_t(a).requires_grad_(True) for a in (xys, conics, opac))
            c1, c2 = _t(colors[:, :32]).requires_grad_(True), _t(colors[:, 32:]).requires_grad_(True)
            P.clear_bin_cache()
            imgs = P.rasterize_segments(xt, _t(depths), _t(radii), ct, _t(nth), ot, h, w,
                                        [(c1, _t(bg[:32])), (c2, _t(bg[32:]))])
            g = torch.Generator(device="cpu").manual_seed(3)
            cots = [torch.randn(i.shape, generator=g).to(DEV) for i in imgs]
            torch.autograd.backward(imgs, cots)
            outs[on] = ([i.detach().clone() for i in imgs], [t.grad.clone() for t in (xt, ct, ot, c1, c2)])
        finally:
            P.USE_QUAD_LISTS = prev
    for a, b in zip(outs[True][0], outs[False][0]):
        assert torch.equal(a, b)
    for a, b in zip(outs[True][1], outs[False][1]):
        scale = float(b.abs().max())
        assert float((a - b).abs().max()) <= 2e-6 * scale + 1e-30


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
def test_pair_backward_survives_garbage_final_idx_and_stale_list_counts(oracle, bad):
    """the pair entries: a garbage final_idx image (both walks), and quad-list counts overwritten with INT_MAX / -1
    (held to each quadrant's capacity): status 0, finite gradients, no fault; a short or misaligned list buffer is
    refused with an error code"""
    pc = PairCall(oracle, 3000, 64, 80, 7, seed=6)
    pc.forward(True)
    v1, v2 = torch.randn(64, 80, 32, device=DEV), torch.randn(64, 80, 7, device=DEV)
    fi_bad = torch.full((64, 80), bad, dtype=torch.int32, device=DEV)
    for lists in (True, False):
        st, rec, col = pc.backward(v1, v2, lists, final_idx=fi_bad)
        assert st == 0
        if bad < 0:
            assert float(rec.abs().max()) == 0.0 and float(col.abs().max()) == 0.0
    ntiles = pc.tile_bins.shape[0]
    pc.ql[:ntiles * 16].view(torch.int32).fill_(bad)
    st, rec, col = pc.backward(v1, v2, True)     # counts held to [0, capacity]: records past the forward's are
    assert st == 0                               # whatever the buffer holds, read in bounds
    pc.forward(True)
    st, rec, col = pc.backward(v1, v2, True)     # alive, and right again with the forward's own lists
    assert st == 0 and bool(torch.isfinite(rec).all()) and bool(torch.isfinite(col).all())
    # refused: buffer one byte short, and a misaligned one
    full = pc.ql
    pc.ql = full[:-1]
    st, _, _ = pc.backward(v1, v2, True, expect_ok=False)
    assert st != 0
    pc.ql = torch.empty(full.numel() + 16, dtype=torch.uint8, device=DEV)[4:]
    st, _, _ = pc.backward(v1, v2, True, expect_ok=False)
    assert st != 0
