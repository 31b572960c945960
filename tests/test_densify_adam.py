"""densify / cull / split + Adam-state surgery + fused Adam.

This row IS pinnable (unlike the rasterizer): the reference does it with plain torch
(nerfstudio/models/gaussian_splatting.py:333-393,402-546; nerfstudio/engine/optimizers.py:158-171 ->
torch.optim.Adam), and torch is importable here.  CPU tests pin the oracle's functions against that
torch behaviour (restated below with torch ops, a few lines each, citing the lines they follow); the
`-m gpu` tests then hold the HIP kernels to the oracle (masks / indices / copied rows bit-exact) and to
torch itself (torch.optim.Adam on the GPU, boolean indexing, torch.cat)."""
import numpy as np
import pytest
import torch

from gaussiangrasper_amd.densify import GROUPS, RefineConfig

REF_GROUPS = {  # nerfstudio/configs/method_configs.py:618-660
    "xyz": dict(lr=1.6e-4, eps=1e-15), "color": dict(lr=5e-4, eps=1e-15), "feature": dict(lr=5e-4, eps=1e-15),
    "opacity": dict(lr=0.05, eps=1e-15), "scaling": dict(lr=0.005, eps=1e-15), "rotation": dict(lr=0.001, eps=1e-15)}


def _ulp_diff(a, b):
    a, b = np.asarray(a, np.float32), np.asarray(b, np.float32)
    ia, ib = a.view(np.int32).astype(np.int64), b.view(np.int32).astype(np.int64)
    ia = np.where(ia < 0, -(ia & 0x7FFFFFFF), ia)
    ib = np.where(ib < 0, -(ib & 0x7FFFFFFF), ib)
    return np.abs(ia - ib)


def _params(n, d=32, seed=0, device="cpu"):
    g = torch.Generator().manual_seed(seed)
    p = {"means": torch.randn(n, 3, generator=g), "scales": torch.randn(n, 3, generator=g) - 4.5,
         "quats": torch.randn(n, 4, generator=g), "opacities": 2 * torch.randn(n, 1, generator=g),
         "colors_all": torch.randn(n, 25, 3, generator=g), "feature": torch.randn(n, d, generator=g)}
    return {k: v.to(device) for k, v in p.items()}


# ------------------------------------------------------------------------------------------------
# torch restatements of the reference lines (test-side; the thing the oracle is pinned to)
# ------------------------------------------------------------------------------------------------
def torch_split_dup(p, split, dup, samps, z, quat_to_rotmat):
    """split_gaussians :504-531 + dup_gaussians :533-546 + the torch.cat of :434-439"""
    q = p["quats"][split] / p["quats"][split].norm(dim=-1, keepdim=True)
    rots = quat_to_rotmat(q.repeat(samps, 1))
    scaled = torch.exp(p["scales"][split].repeat(samps, 1)) * z
    new_means = torch.bmm(rots, scaled[..., None]).squeeze(-1) + p["means"][split].repeat(samps, 1)
    shrunk = torch.log(torch.exp(p["scales"][split]) / 1.6)
    scales_old = p["scales"].clone()
    scales_old[split] = shrunk
    rep = lambda t: t[split].repeat(samps, *([1] * (t.dim() - 1)))
    # dup_gaussians (:541) copies self.scales AFTER split_gaussians shrank the split rows in place (:524-526)
    return {"means": torch.cat([p["means"], new_means, p["means"][dup]]),
            "scales": torch.cat([scales_old, shrunk.repeat(samps, 1), scales_old[dup]]),
            "quats": torch.cat([p["quats"], rep(p["quats"]), p["quats"][dup]]),
            "opacities": torch.cat([p["opacities"], rep(p["opacities"]), p["opacities"][dup]]),
            "colors_all": torch.cat([p["colors_all"], rep(p["colors_all"]), p["colors_all"][dup]]),
            "feature": torch.cat([p["feature"], rep(p["feature"]), p["feature"][dup]])}


def torch_dup_in_optim(m, split, dup, samps):
    """dup_in_optim :352-371 applied for the splits (n=samps) and then the dups (n=1)"""
    z = lambda mask, n: torch.zeros_like(m[mask]).repeat(n, *([1] * (m.dim() - 1)))
    return torch.cat([m, z(split, samps), z(dup, 1)])


def torch_stats(norm, counts, size, xys_grad, radii, max_dim):
    """after_train :373-393"""
    vis = radii > 0
    grads = xys_grad.norm(dim=-1)
    if norm is None:
        norm, counts = grads.clone(), torch.ones_like(grads)
    else:
        counts, norm = counts.clone(), norm.clone()
        counts[vis] = counts[vis] + 1
        norm[vis] = grads[vis] + norm[vis]
    size = torch.zeros_like(grads) if size is None else size.clone()
    size[vis] = torch.maximum(size[vis], radii[vis] / float(max_dim))
    return norm, counts, size


def torch_masks(norm, counts, size, scales, max_dim, cfg, step):
    """refinement_after :412-421,:430-431"""
    avg = (norm / counts) * 0.5 * max_dim
    high = avg > cfg.densify_grad_thresh
    big = scales.exp().max(dim=-1).values > cfg.densify_size_thresh
    splits = big.clone()
    if step < cfg.stop_screen_size_at:
        splits |= size > cfg.split_screen_size
    splits = splits & high
    # the reference's statement order: split_gaussians (:423-429) has shrunk self.scales[splits] in place (:524-526)
    # when `dups = self.scales.exp().max(dim=-1).values <= thresh` (:430) is evaluated
    scales = scales.clone()
    scales[splits] = torch.log(torch.exp(scales[splits]) / 1.6)
    dups = (scales.exp().max(dim=-1).values <= cfg.densify_size_thresh) & high
    return splits, dups


def torch_cull(opac, scales, size, cfg, step):
    """cull_gaussians :485-496"""
    culls = (torch.sigmoid(opac) < cfg.cull_alpha_thresh).squeeze(-1)
    if step > cfg.refine_every * cfg.reset_alpha_every:
        culls = culls | (torch.exp(scales).max(dim=-1).values > cfg.cull_scale_thresh)
        if step < cfg.stop_screen_size_at:
            culls = culls | (size > cfg.cull_screen_size)
    return culls


def _away(x, thr, rel=1e-4):
    """move values that sit within rel of a threshold off it (exp/sigmoid differ in the last bit
    between libm, torch and the GPU; the masks are compared bit for bit)"""
    x = x.clone()
    close = (x - thr).abs() <= rel * abs(thr)
    x[close] = thr * (1 + 4 * rel)
    return x


# ------------------------------------------------------------------------------------------------
# CPU: the oracle against torch
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("group", list(REF_GROUPS))
def test_oracle_adam_matches_torch_optim_adam(oracle, group):
    hp = REF_GROUPS[group]
    g = torch.Generator().manual_seed(11)
    p0 = torch.randn(4099, generator=g)
    ref = torch.nn.Parameter(p0.clone())
    opt = torch.optim.Adam([ref], lr=hp["lr"], eps=hp["eps"], foreach=False)
    p, m, v = p0.numpy().copy(), np.zeros(4099, np.float32), np.zeros(4099, np.float32)
    for step in range(1, 8):
        grad = torch.randn(4099, generator=g) * (10.0 ** float(torch.randint(-6, 1, (1,), generator=g)))
        ref.grad = grad.clone()
        opt.step()
        p, m, v = oracle.adam_step(p, grad.numpy(), m, v, lr=hp["lr"], eps=hp["eps"], step=step)
        st = opt.state[ref]
        assert _ulp_diff(m, st["exp_avg"].numpy()).max() <= 2, step
        assert _ulp_diff(v, st["exp_avg_sq"].numpy()).max() <= 2, step
        # the update is lr-sized: compare the parameter at its own ulp scale
        assert np.abs(p - ref.detach().numpy()).max() <= 4e-7 * max(1.0, np.abs(p).max()) * step


def test_oracle_adam_weight_decay_and_f64(oracle):
    g = torch.Generator().manual_seed(2)
    p0, grad = torch.randn(257, generator=g, dtype=torch.float64), torch.randn(257, generator=g, dtype=torch.float64)
    ref = torch.nn.Parameter(p0.clone())
    opt = torch.optim.Adam([ref], lr=1e-2, eps=1e-8, weight_decay=0.1, betas=(0.8, 0.95), foreach=False)
    ref.grad = grad.clone()
    opt.step()
    p, m, v = oracle.adam_step(p0.numpy(), grad.numpy(), np.zeros(257), np.zeros(257), lr=1e-2, beta1=0.8,
                               beta2=0.95, eps=1e-8, weight_decay=0.1, step=1, dtype=np.float64)
    assert np.allclose(p, ref.detach().numpy(), rtol=1e-13, atol=1e-15)
    assert np.allclose(m, opt.state[ref]["exp_avg"].numpy(), rtol=1e-13)


def test_oracle_compaction_and_scan_match_torch_indexing(oracle):
    g = torch.Generator().manual_seed(3)
    for n in (1, 5, 1024, 3001):
        mask = torch.rand(n, generator=g) < 0.37
        ranks, total = oracle.mask_scan(mask.numpy())
        assert total == int(mask.sum())
        assert np.array_equal(ranks, (torch.cumsum(mask.int(), 0) - mask.int()).numpy())
        r2, t2 = oracle.mask_scan(mask.numpy(), invert=True)
        assert t2 == n - total
        for shape in ((n, 3), (n, 25, 3), (n, 1), (n,)):
            t = torch.randn(*shape, generator=g)
            out = oracle.compact_rows(mask.numpy(), t.numpy())
            assert np.array_equal(out, t[~mask].numpy())


@pytest.mark.parametrize("overlap", [False, True])
def test_oracle_split_dup_matches_the_reference_torch_code(oracle, overlap):
    """overlap: Gaussians in BOTH masks (split for their screen size or with thresh < size <= 1.6 thresh, then
    small enough after the in-place shrink to be duplicated as well): their dup row carries the shrunk scale"""
    import oracle_ops
    n, samps = 700, 2
    p = _params(n, d=8, seed=4)
    g = torch.Generator().manual_seed(5)
    split = torch.rand(n, generator=g) < 0.2
    dup = (torch.rand(n, generator=g) < 0.3)
    if not overlap:
        dup = dup & ~split
    else:
        assert int((dup & split).sum()) > 10
    z = torch.randn(samps * int(split.sum()), 3, generator=g)
    want = torch_split_dup(p, split, dup, samps, z, oracle_ops.quat_to_rotmat)
    kinds = {"means": oracle.ROWS_MEANS, "scales": oracle.ROWS_SCALES}
    for name, t in p.items():
        got = oracle.densify_rows(t.numpy(), kinds.get(name, oracle.ROWS_COPY), split.numpy(), dup.numpy(),
                                  samps, z.numpy(), 1.6, p["means"].numpy(), p["scales"].numpy(),
                                  p["quats"].numpy())
        w = want[name].numpy()
        assert got.shape == w.shape, name
        if name in ("means", "scales"):
            # exp / log / the 3x3 product differ in the last bits between libm and torch
            assert np.allclose(got, w, rtol=2e-6, atol=2e-6), name
            untouched = np.ones(len(w), bool)
            untouched[n:n + samps * int(split.sum())] = False
            if name == "scales":
                untouched[:n][split.numpy()] = False
            assert np.array_equal(got[untouched], w[untouched]), name
        else:
            assert np.array_equal(got, w), name
    m = torch.randn(n, 25, 3, generator=g)
    got = oracle.densify_rows(m.numpy(), oracle.ROWS_ZERO_NEW, split.numpy(), dup.numpy(), samps, z.numpy(),
                              1.6, p["means"].numpy(), p["scales"].numpy(), p["quats"].numpy())
    assert np.array_equal(got, torch_dup_in_optim(m, split, dup, samps).numpy())


def test_oracle_stats_and_masks_match_the_reference_torch_code(oracle):
    cfg = RefineConfig()
    n, max_dim = 5000, 1600
    g = torch.Generator().manual_seed(6)
    norm = counts = size = None
    on = oc = osz = None
    for it in range(3):
        xg = torch.randn(n, 2, generator=g) * 1e-6
        radii = torch.randint(-2, 400, (n,), generator=g).clamp(min=0).int()
        norm, counts, size = torch_stats(norm, counts, size, xg, radii, max_dim)
        on, oc, osz = oracle.densify_stats(xg.numpy(), radii.numpy(), max_dim, it == 0, on, oc, osz)
        assert np.array_equal(oc, counts.numpy()) and np.array_equal(osz, size.numpy())
        assert _ulp_diff(on, norm.numpy()).max() <= 1 + it  # sqrt(x^2+y^2): torch sums the squares its own way
    scales = _away(torch.randn(n, 3, generator=g) * 1.5 - 4.6, float(np.log(cfg.densify_size_thresh)))
    scales = _away(scales, float(np.log(1.6 * cfg.densify_size_thresh)))   # the duplicate test of a split Gaussian
    scales = _away(scales, float(np.log(cfg.cull_scale_thresh)))
    opac = _away(2 * torch.randn(n, 1, generator=g), float(np.log(0.1 / 0.9)))
    norm_t = torch.from_numpy(on)
    for step in (1000, 3500, 5000):
        s_t, d_t = torch_masks(norm_t, counts, size, scales, max_dim, cfg, step)
        s_o, d_o = oracle.densify_masks(on, oc, osz, scales.numpy(), max_dim, cfg.densify_grad_thresh,
                                        cfg.densify_size_thresh, cfg.split_screen_size,
                                        step < cfg.stop_screen_size_at)
        assert np.array_equal(s_o, s_t.numpy()) and np.array_equal(d_o, d_t.numpy())
        assert s_o.any() and d_o.any()
        # both masks at once, the reference's statement order (:423-431): a size in (thresh, 1.6 thresh] ...
        smax = scales.exp().max(dim=-1).values.numpy()
        mid = (smax > cfg.densify_size_thresh) & (smax <= 1.6 * cfg.densify_size_thresh * (1 - 1e-3))
        assert (s_o & d_o & mid).any() and np.array_equal((s_o & mid), (d_o & mid))
        # ... and a small Gaussian split for its screen size
        if step < cfg.stop_screen_size_at:
            small = smax <= cfg.densify_size_thresh
            assert (s_o & d_o & small).any()
        c_t = torch_cull(opac, scales, size, cfg, step)
        c_o = oracle.cull_mask(opac.numpy(), scales.numpy(), osz, cfg.cull_alpha_thresh, cfg.cull_scale_thresh,
                               cfg.cull_screen_size, step > cfg.refine_every * cfg.reset_alpha_every,
                               step < cfg.stop_screen_size_at)
        assert np.array_equal(c_o, c_t.numpy()) and c_o.any() and not c_o.all()


def test_abi_declares_and_types_the_f3_entry_points():
    from gaussiangrasper_amd import _lib
    lib = _lib.load()
    for name in ("gg_rows_workspace", "gg_mask_scan", "gg_compact_rows", "gg_densify_rows", "gg_densify_stats",
                 "gg_densify_masks", "gg_cull_mask", "gg_adam_step"):
        assert name in _lib.SIGNATURES and hasattr(lib, name)
    import ctypes as C
    assert C.sizeof(_lib.RowArray) == 24 and C.sizeof(_lib.AdamGroup) == 88
    assert lib.gg_rows_workspace(0) >= 32 and lib.gg_rows_workspace(1 << 20) >= 32 + 8 * 1024


def test_fused_adam_and_refiner_refuse_cpu_tensors():
    from gaussiangrasper_amd.densify import compact
    from gaussiangrasper_amd.optim import FusedAdam
    p = torch.nn.Parameter(torch.zeros(8))
    opt = FusedAdam([p], lr=1e-3)
    p.grad = torch.ones(8)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        opt.step()
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        compact([torch.zeros(4, 3)], torch.zeros(4, dtype=torch.bool))
    with pytest.raises(NotImplementedError):
        FusedAdam([p], amsgrad=True)


# ------------------------------------------------------------------------------------------------
# GPU: the HIP kernels against the oracle and against torch on the same device
# ------------------------------------------------------------------------------------------------
DEV = "cuda:0"
gpu = pytest.mark.gpu


@gpu
@pytest.mark.parametrize("n", [1, 3, 1023, 1024, 1025, 50_000, 3_000_001])
def test_gpu_compaction_bitexact(oracle, n):
    from gaussiangrasper_amd.densify import compact, mask_scan
    g = torch.Generator().manual_seed(n)
    for frac in (0.0, 0.41, 1.0):
        mask = torch.rand(n, generator=g) < frac
        md = mask.to(DEV)
        ranks, total = mask_scan(md)
        assert total == int(mask.sum())
        assert torch.equal(ranks.cpu(), (torch.cumsum(mask.int(), 0) - mask.int()).int())
        arrays = [torch.randn(n, 3, generator=g), torch.randn(n, 25, 3, generator=g), torch.randn(n, 1, generator=g),
                  torch.randn(n, 32, generator=g)] if n < 1_000_000 else [torch.randn(n, 3, generator=g),
                                                                           torch.randn(n, 32, generator=g)]
        outs = compact([a.to(DEV) for a in arrays], md)
        for a, o in zip(arrays, outs):
            assert torch.equal(o.cpu(), a[~mask])
        if n <= 50_000:
            assert np.array_equal(outs[1].cpu().numpy(), oracle.compact_rows(mask.numpy(), arrays[1].numpy()))


@gpu
def test_gpu_compaction_of_eighteen_arrays_in_one_launch():
    """the 6 parameters + 12 Adam moments of a cull (ref :497-502 + :333-350) through ONE gg_compact_rows"""
    from gaussiangrasper_amd.densify import compact
    n = 20_000
    p = _params(n, seed=8)
    arrays = []
    for t in p.values():
        arrays += [t, torch.randn_like(t), torch.rand_like(t)]
    mask = torch.rand(n) < 0.3
    outs = compact([a.to(DEV) for a in arrays], mask.to(DEV))
    assert len(outs) == 18
    for a, o in zip(arrays, outs):
        assert torch.equal(o.cpu(), a[~mask])


@gpu
@pytest.mark.parametrize("n,samps,overlap", [(700, 2, False), (40_000, 2, True), (3000, 3, True)])
def test_gpu_split_dup_vs_oracle_and_torch(oracle, n, samps, overlap):
    """overlap: Gaussians in both masks (the reference's statement order allows it, :423-431): their duplicate row
    carries the scale split_gaussians has already shrunk"""
    from gaussiangrasper_amd import _lib
    from gaussiangrasper_amd import ops as P
    from gaussiangrasper_amd.densify import append_rows
    p = _params(n, d=32, seed=n)
    g = torch.Generator().manual_seed(n + 1)
    split = torch.rand(n, generator=g) < 0.15
    dup = (torch.rand(n, generator=g) < 0.25)
    if not overlap:
        dup = dup & ~split
    else:
        assert int((dup & split).sum()) > 10
    z = torch.randn(samps * int(split.sum()), 3, generator=g)
    kinds = {"means": _lib.ROWS_MEANS, "scales": _lib.ROWS_SCALES}
    names = list(p)
    moment = torch.randn(n, 25, 3, generator=g)
    arrays = [(p[k].to(DEV), kinds.get(k, _lib.ROWS_COPY)) for k in names] + [(moment.to(DEV), _lib.ROWS_ZERO_NEW)]
    outs, ns, nd, used = append_rows(arrays, split.to(DEV), dup.to(DEV), samps, z.to(DEV), p["means"].to(DEV),
                                     p["scales"].to(DEV), p["quats"].to(DEV))
    assert ns == int(split.sum()) and nd == int(dup.sum())
    pd = {k: v.to(DEV) for k, v in p.items()}
    want_t = torch_split_dup(pd, split.to(DEV), dup.to(DEV), samps, z.to(DEV), P.quat_to_rotmat)
    for k, o in zip(names, outs):
        want_o = oracle.densify_rows(p[k].numpy(), kinds.get(k, oracle.ROWS_COPY), split.numpy(), dup.numpy(),
                                     samps, z.numpy(), 1.6, p["means"].numpy(), p["scales"].numpy(),
                                     p["quats"].numpy())
        got = o.cpu().numpy()
        if k in ("means", "scales"):
            assert np.allclose(got, want_o, rtol=2e-6, atol=2e-6), k           # expf / logf: ocml vs glibc
            assert np.allclose(got, want_t[k].cpu().numpy(), rtol=2e-6, atol=2e-6), k
            assert np.array_equal(got[:n][~split.numpy()], p[k].numpy()[~split.numpy()])
            only_dup = (dup & ~split)[dup].numpy()      # rows that are duplicated without having been split
            assert np.array_equal(got[n + samps * ns:][only_dup], p[k][dup].numpy()[only_dup])
        else:
            assert np.array_equal(got, want_o), k
            assert torch.equal(o, want_t[k]), k
    assert torch.equal(outs[-1].cpu(), torch_dup_in_optim(moment, split, dup, samps))


@gpu
def test_gpu_stats_masks_bitexact_vs_oracle(oracle):
    from gaussiangrasper_amd.densify import Refiner
    cfg = RefineConfig()
    n, size_hw = 30_000, (1200, 1600)
    p = _params(n, seed=21)
    g = torch.Generator().manual_seed(22)
    p["scales"] = _away(_away(_away(torch.randn(n, 3, generator=g) * 1.5 - 4.6, float(np.log(cfg.densify_size_thresh))),
                              float(np.log(1.6 * cfg.densify_size_thresh))),
                        float(np.log(cfg.cull_scale_thresh)))
    p["opacities"] = _away(2 * torch.randn(n, 1, generator=g), float(np.log(0.1 / 0.9)))
    ref = Refiner({k: v.to(DEV) for k, v in p.items()}, {}, cfg)
    on = oc = osz = None
    for it in range(3):
        xg = torch.randn(n, 2, generator=g) * 1e-6
        radii = torch.randint(-2, 400, (n,), generator=g).clamp(min=0).int()
        ref.after_train(xg.to(DEV), radii.to(DEV), size_hw)
        on, oc, osz = oracle.densify_stats(xg.numpy(), radii.numpy(), 1600, it == 0, on, oc, osz)
    assert np.array_equal(ref.xys_grad_norm.cpu().numpy().view(np.uint32), on.view(np.uint32))
    assert np.array_equal(ref.vis_counts.cpu().numpy(), oc) and np.array_equal(ref.max_2Dsize.cpu().numpy(), osz)
    for step in (1000, 3500, 5000):
        ref.step = step
        s, d = ref.densify_masks()
        s_o, d_o = oracle.densify_masks(on, oc, osz, p["scales"].numpy(), 1600, cfg.densify_grad_thresh,
                                        cfg.densify_size_thresh, cfg.split_screen_size,
                                        step < cfg.stop_screen_size_at)
        assert np.array_equal(s.cpu().numpy().astype(bool), s_o) and np.array_equal(d.cpu().numpy().astype(bool), d_o)
        c = ref.cull_mask()
        c_o = oracle.cull_mask(p["opacities"].numpy(), p["scales"].numpy(), osz, cfg.cull_alpha_thresh,
                               cfg.cull_scale_thresh, cfg.cull_screen_size,
                               step > cfg.refine_every * cfg.reset_alpha_every, step < cfg.stop_screen_size_at)
        assert np.array_equal(c.cpu().numpy().astype(bool), c_o)


@gpu
@pytest.mark.parametrize("numel", [1, 5, 4096, 1_000_003])
def test_gpu_adam_bitexact_vs_oracle(oracle, numel):
    """gg_adam_step through the C ABI: parameters and both moments bit for bit the oracle's"""
    import ctypes as C
    from gaussiangrasper_amd import _lib
    from gaussiangrasper_amd import ops as P
    lib = _lib.load()
    g = torch.Generator().manual_seed(numel)
    p = torch.randn(numel, generator=g)
    m = torch.randn(numel, generator=g) * 1e-3
    v = torch.rand(numel, generator=g) * 1e-5
    grad = torch.randn(numel, generator=g) * 1e-2
    pd, md, vd, gd = (t.clone().to(DEV) for t in (p, m, v, grad))
    for step, (lr, eps, wd) in enumerate([(1.6e-4, 1e-15, 0.0), (0.05, 1e-15, 0.0), (1e-3, 1e-8, 0.01)], start=3):
        arr = (_lib.AdamGroup * 1)(_lib.AdamGroup(pd.data_ptr(), gd.data_ptr(), md.data_ptr(), vd.data_ptr(),
                                                 numel, lr, 0.9, 0.999, eps, wd, step))
        _lib.check(lib.gg_adam_step(1, arr, 0, P._stream(pd.device)), "gg_adam_step")
        pn, mn, vn = oracle.adam_step(p.numpy(), grad.numpy(), m.numpy(), v.numpy(), lr=lr, eps=eps,
                                      weight_decay=wd, step=step)
        for name, a, b in (("exp_avg", md, mn), ("exp_avg_sq", vd, vn), ("param", pd, pn)):
            assert np.array_equal(a.cpu().numpy().view(np.uint32), b.view(np.uint32)), (name, step)
        p, m, v = torch.from_numpy(pn), torch.from_numpy(mn), torch.from_numpy(vn)
    # zero_grad in the same pass
    arr = (_lib.AdamGroup * 1)(_lib.AdamGroup(pd.data_ptr(), gd.data_ptr(), md.data_ptr(), vd.data_ptr(),
                                             numel, 1e-3, 0.9, 0.999, 1e-8, 0.0, 9))
    _lib.check(lib.gg_adam_step(1, arr, 1, P._stream(pd.device)), "gg_adam_step")
    assert not gd.any()


@gpu
def test_gpu_fused_adam_tracks_torch_adam_over_the_reference_groups():
    """six groups with the reference's hyper-parameters, ONE launch per step, against six
    torch.optim.Adam instances on the same device"""
    from gaussiangrasper_amd.optim import FusedAdam, fused_step
    n = 10_000
    base = _params(n, seed=31)
    mine = {k: torch.nn.Parameter(v.clone().to(DEV)) for k, v in base.items()}
    ref = {k: torch.nn.Parameter(v.clone().to(DEV)) for k, v in base.items()}
    o_mine = {gname: FusedAdam([mine[attr]], **REF_GROUPS[gname]) for gname, attr in GROUPS.items()}
    o_ref = {gname: torch.optim.Adam([ref[attr]], **REF_GROUPS[gname]) for gname, attr in GROUPS.items()}
    g = torch.Generator().manual_seed(32)
    for step in range(1, 6):
        for k in base:
            grad = (torch.randn(base[k].shape, generator=g) * 1e-3).to(DEV)
            mine[k].grad, ref[k].grad = grad.clone(), grad.clone()
        fused_step(list(o_mine.values()))
        for o in o_ref.values():
            o.step()
    for gname, attr in GROUPS.items():
        sm, sr = o_mine[gname].state[mine[attr]], o_ref[gname].state[ref[attr]]
        assert float(sm["step"]) == 5 == float(sr["step"])
        assert torch.allclose(sm["exp_avg"], sr["exp_avg"], rtol=2e-6, atol=1e-12)
        assert torch.allclose(sm["exp_avg_sq"], sr["exp_avg_sq"], rtol=2e-6, atol=1e-18)
        assert torch.allclose(mine[attr], ref[attr], rtol=0, atol=5e-6 * REF_GROUPS[gname]["lr"] / 1e-4 + 2e-6)
    # state_dict layout is torch.optim.Adam's: a torch optimizer loads it
    sd = o_mine["xyz"].state_dict()
    assert set(sd["state"][0]) == {"step", "exp_avg", "exp_avg_sq"}
    o_ref["xyz"].load_state_dict(sd)


@gpu
def test_gpu_refinement_after_end_to_end_against_the_torch_restatement():
    """Refiner.refinement_after (densify + cull + optimizer surgery + GradBucket re-aliasing) against
    the reference's torch sequence (:402-473) on the same device, same N(0,1) draws"""
    from gaussiangrasper_amd import ops as P
    from gaussiangrasper_amd.densify import Refiner
    from gaussiangrasper_amd.dist import GradBucket
    from gaussiangrasper_amd.optim import FusedAdam
    cfg = RefineConfig()
    n, hw = 12_000, (1200, 1600)
    base = _params(n, seed=41)
    g = torch.Generator().manual_seed(42)
    base["scales"] = _away(_away(_away(torch.randn(n, 3, generator=g) * 1.2 - 4.6, float(np.log(cfg.densify_size_thresh))),
                                 float(np.log(1.6 * cfg.densify_size_thresh))),
                           float(np.log(cfg.cull_scale_thresh)))
    base["opacities"] = _away(2 * torch.randn(n, 1, generator=g), float(np.log(0.1 / 0.9)))
    params = {k: torch.nn.Parameter(v.clone().to(DEV)) for k, v in base.items()}
    opts = {gname: FusedAdam([params[attr]], **REF_GROUPS[gname]) for gname, attr in GROUPS.items()}
    order = ("means", "scales", "quats", "opacities", "colors_all", "feature")
    bucket = GradBucket([params[k] for k in order])
    for k in order:                     # one optimizer step so that every group has moments
        params[k].grad.copy_((torch.randn(base[k].shape, generator=g) * 1e-3).to(DEV))
    for o in opts.values():
        o.step()
    moments0 = {attr: {k: opts[gname].state[params[attr]][k].clone() for k in ("exp_avg", "exp_avg_sq")}
                for gname, attr in GROUPS.items()}
    p0 = {k: v.detach().clone() for k, v in params.items()}
    ref = Refiner(params, opts, cfg, num_train_data=50, bucket=bucket)
    xg = (torch.randn(n, 2, generator=g) * 2e-6).to(DEV)
    radii = torch.randint(-2, 300, (n,), generator=g).clamp(min=0).int().to(DEV)
    ref.after_train(xg, radii, hw)
    step = 3400                         # in the densify window: 3400 % 3000 = 400 > 50 + 100
    norm, counts, size = torch_stats(None, None, None, xg, radii, 1600)
    split, dup = torch_masks(norm, counts, size, p0["scales"], 1600, cfg, step)
    z = torch.randn(cfg.n_split_samples * int(split.sum()), 3, generator=g).to(DEV)
    info = ref.refinement_after(step, samples=z)
    want = torch_split_dup(p0, split, dup, cfg.n_split_samples, z, P.quat_to_rotmat)
    size2 = torch.cat([size, size.new_zeros(want["means"].shape[0] - n)])
    culls = torch_cull(want["opacities"], want["scales"], size2, cfg, step)
    assert info == {"split": int(split.sum()), "dup": int(dup.sum()), "culled": int(culls.sum()), "opacity_reset": 0}
    assert info["split"] > 0 and info["dup"] > 0 and info["culled"] > 0
    n_new = int((~culls).sum())
    assert ref.num_points == n_new
    for k in order:
        got, w = ref.params[k].detach(), want[k][~culls]
        assert got.shape == w.shape, k
        if k in ("means", "scales"):
            assert torch.allclose(got, w, rtol=2e-6, atol=2e-6), k
        else:
            assert torch.equal(got, w), k
    for gname, attr in GROUPS.items():
        opt = opts[gname]
        assert opt.param_groups[0]["params"] == [ref.params[attr]] and len(opt.state) == 1
        st = opt.state[ref.params[attr]]
        for key in ("exp_avg", "exp_avg_sq"):
            w = torch_dup_in_optim(moments0[attr][key], split, dup, cfg.n_split_samples)[~culls]
            assert torch.equal(st[key], w), (gname, key)
    # the bucket was re-aliased: gradients of the NEW parameters land in it, and a step works
    assert bucket.payload == n_new * 118 and bucket.nbytes == n_new * 472
    for k in order:
        assert ref.params[k].grad.data_ptr() == bucket.slices[order.index(k)].data_ptr()
        ref.params[k].grad.fill_(1e-3)
    for o in opts.values():
        o.step()
    assert ref.xys_grad_norm is None and ref.max_2Dsize is None
    # opacity reset step (:459-470)
    ref.after_train(torch.zeros(n_new, 2, device=DEV), torch.ones(n_new, dtype=torch.int32, device=DEV), hw)
    info = ref.refinement_after(3100)
    assert info["opacity_reset"] == 1 and info["split"] == 0 and info["culled"] == 0
    val = torch.logit(torch.tensor(cfg.cull_alpha_thresh * 0.8)).item()
    assert torch.all(ref.params["opacities"] == val)
    st = opts["opacity"].state[ref.params["opacities"]]
    assert not st["exp_avg"].any() and not st["exp_avg_sq"].any()


@gpu
def test_plugin_class_trains_with_the_fused_optimizer_side():
    """What train.sh gets from the plugin (fused_training): the trainer's own loop (trainer.py:459-498 —
    zero_grad, model(camera), loss.backward(), optimizer steps, then the AFTER_TRAIN_ITERATION callbacks in the order
    get_training_callbacks registers them, gaussian_splatting.py:548-571) on the plugin's model class, with
    optim.FusedAdam for the six Gaussian groups (what plugin._spec puts into the config) and its after_train /
    refinement_after on densify.Refiner.  250 iterations across three refinements: the class swaps the new Parameters
    into the model AND into the optimizers with moments of matching shape, statistics restart, rendering continues at
    the new N; and the first refinement equals the reference's torch sequence on the same state."""
    import types
    from gaussiangrasper_amd import ops as P
    from gaussiangrasper_amd.camera import ring_cameras
    from gaussiangrasper_amd.optim import FusedAdam
    from gaussiangrasper_amd.plugin import make_fused_model_class
    from gaussiangrasper_amd.scene import make_scene
    from gaussiangrasper_amd.stub import StubCameras, StubGaussianSplattingModel, default_config
    torch.manual_seed(0)
    n, h, w, nviews = 4000, 96, 128, 6
    sc = make_scene(n, feature_dim=32, config_index=2)
    sc.scales.add_(1.0)
    cfg = default_config(num_downscales=0, densify_grad_thresh=2e-7)     # full resolution; this toy scene's gradients are small
    Model = make_fused_model_class(StubGaussianSplattingModel, fused_training=True)
    m = Model(sc, config=cfg, num_train_data=nviews, step=500).to(DEV)
    m.train()
    groups = m.get_gaussian_param_groups()
    optimizers = types.SimpleNamespace(
        optimizers={g: FusedAdam(ps, **REF_GROUPS[g]) for g, ps in groups.items()},
        parameters={g: list(ps) for g, ps in groups.items()})
    cams = [StubCameras.from_view(v, device=DEV, cam_idx=i) for i, v in enumerate(ring_cameras(nviews, h, w))]
    gen = torch.Generator().manual_seed(1)
    targets = [torch.rand(h, w, 3, generator=gen).to(DEV) for _ in range(nviews)]
    counts, first_checked = [], False
    for step in range(500, 750):
        m.step_cb(step)
        for o in optimizers.optimizers.values():
            o.zero_grad(set_to_none=True)
        out = m(cams[step % nviews])
        loss = (out["rgb"] - targets[step % nviews]).abs().mean() + 0.1 * out["feature"].pow(2).mean() \
            + 0.01 * out["depth"].mean() + 0.01 * out["normal"].pow(2).mean()
        loss.backward()
        for o in optimizers.optimizers.values():
            o.step()
        m.after_train(step)
        if step % cfg.refine_every == 0:
            if not first_checked and step % (cfg.reset_alpha_every * cfg.refine_every) > nviews + cfg.refine_every:
                # the state the refinement starts from, for the reference's torch sequence (:412-431, :485-496)
                norm, cnt, size = m.xys_grad_norm.clone(), m.vis_counts.clone(), m.max_2Dsize.clone()
                p0 = {a: getattr(m, a).detach().clone() for a in GROUPS.values()}
                split, dup = torch_masks(norm, cnt, size, p0["scales"], max(h, w), cfg, step)
            before = m.num_points
            info = m.refinement_after(optimizers, step)
            if not first_checked and info["split"] + info["dup"] > 0:
                first_checked = True
                assert info["split"] == int(split.sum()) and info["dup"] == int(dup.sum())
                assert m.num_points == before + cfg.n_split_samples * info["split"] + info["dup"] - info["culled"]
            counts.append(m.num_points)
            # the wiring: model attributes, optimizer param_groups and moments, the trainer's parameter table
            for g, a in GROUPS.items():
                p = getattr(m, a)
                opt = optimizers.optimizers[g]
                assert opt.param_groups[0]["params"][0] is p and optimizers.parameters[g][0] is p
                assert p.shape[0] == m.num_points and p.is_leaf and p.requires_grad
                st = opt.state[p]
                assert st["exp_avg"].shape == p.shape == st["exp_avg_sq"].shape
            assert m.xys_grad_norm is None and m.vis_counts is None and m.max_2Dsize is None
    assert first_checked and len(counts) == 3 and counts[-1] != n     # refinements at steps 500, 600, 700
    assert all(bool(torch.isfinite(getattr(m, a)).all()) for a in GROUPS.values())
    assert m.xys.shape[0] == m.num_points == m.radii.shape[0]
    P.clear_bin_cache()
