"""gg_shade_tail (SH + 0.5 + clamp + rgb | depth | normal packing, reference gaussian_splatting.py:730-731, :765, :779):
the oracle restatement against torch autograd of clamp / cat applied to the oracle's own SH (CPU), and the HIP kernels
against the oracle (GPU, bit-exact both ways)."""
import numpy as np
import pytest
import torch

from gaussiangrasper_amd.ops import num_sh_bases


def _inputs(n, k, seed=0):
    rng = np.random.default_rng(seed)
    viewdirs = rng.standard_normal((n, 3)).astype(np.float32)
    viewdirs /= np.linalg.norm(viewdirs, axis=1, keepdims=True)
    coeffs = (rng.standard_normal((n, k, 3)) * 0.6).astype(np.float32)     # plenty of values outside [-0.5, 0.5]
    depths = rng.uniform(0.5, 9.0, n).astype(np.float32)
    normals = rng.standard_normal((n, 3)).astype(np.float32)
    return viewdirs, coeffs, depths, normals


@pytest.mark.parametrize("k,deg", [(25, 4), (25, 2), (16, 3), (9, 2), (4, 1), (1, 0)])
def test_oracle_shade_tail_matches_torch_autograd_of_the_callers_ops(oracle, k, deg):
    n = 777
    viewdirs, coeffs, depths, normals = _inputs(n, k)
    tail, mask = oracle.shade_tail_fwd(deg, viewdirs, coeffs, depths, normals)
    rgb = torch.from_numpy(oracle.sh_fwd(deg, viewdirs, coeffs)).requires_grad_(True)
    d, nr = torch.from_numpy(depths).requires_grad_(True), torch.from_numpy(normals).requires_grad_(True)
    ref = torch.cat([torch.clamp(rgb + 0.5, 0.0, 1.0), d[:, None], nr], dim=1)
    np.testing.assert_array_equal(tail, ref.detach().numpy())
    # cotangent rows inside a wider record, as the blend backward hands them over
    rec = np.random.default_rng(3).standard_normal((n, 13)).astype(np.float32)
    v_tail = rec[:, 6:]
    ref.backward(torch.from_numpy(np.ascontiguousarray(v_tail)))
    v_sh_ref = oracle.sh_bwd(deg, k, viewdirs, rgb.grad.numpy())
    wide = np.zeros((n, 13), np.float32)
    wide[:, :7] = v_tail
    v_sh, v_d, v_n = oracle.shade_tail_bwd(deg, k, viewdirs, wide, mask)
    np.testing.assert_array_equal(v_sh, v_sh_ref)
    np.testing.assert_array_equal(v_d, d.grad.numpy())
    np.testing.assert_array_equal(v_n, nr.grad.numpy())
    assert 0 < int((mask != 7).sum()) < n          # some colours clamp, some do not
    # accumulation adds to what is there
    prior = np.random.default_rng(4).standard_normal((n, k, 3)).astype(np.float32)
    v_acc, _, _ = oracle.shade_tail_bwd(deg, k, viewdirs, wide, mask, v_coeffs_in=prior)
    np.testing.assert_array_equal(v_acc, prior + v_sh_ref)


def test_clamp_boundaries_pass_the_gradient_like_torch(oracle):
    # x = sh + 0.5 exactly 0 or 1: torch.clamp's backward passes the gradient (min <= x <= max)
    viewdirs = np.array([[0.0, 0.0, 1.0]] * 4, np.float32)
    c0 = 0.28209479177387814
    coeffs = np.zeros((4, 1, 3), np.float32)
    coeffs[0, 0] = -0.5 / c0          # x ~ 0
    coeffs[1, 0] = 0.5 / c0           # x ~ 1
    coeffs[2, 0] = -2.0               # below
    coeffs[3, 0] = 3.0                # above
    tail, mask = oracle.shade_tail_fwd(0, viewdirs, coeffs, np.ones(4, np.float32), np.zeros((4, 3), np.float32))
    x = torch.from_numpy(oracle.sh_fwd(0, viewdirs, coeffs)).requires_grad_(True)
    y = torch.clamp(x + 0.5, 0.0, 1.0)
    y.backward(torch.ones_like(y))
    np.testing.assert_array_equal(tail[:, :3], y.detach().numpy())
    passes = (x.grad.numpy() != 0)
    np.testing.assert_array_equal(((mask[:, None] >> np.arange(3)) & 1).astype(bool), passes)
    assert not passes[2].any() and not passes[3].any()


@pytest.mark.gpu
@pytest.mark.parametrize("n,k,deg", [(1, 25, 4), (63, 25, 4), (1000, 25, 4), (50000, 25, 4), (5000, 25, 2), (3000, 16, 3),
                                     (3000, 9, 2), (3000, 4, 1), (3000, 1, 0)])
def test_hip_shade_tail_bitexact_vs_oracle(oracle, n, k, deg):
    from gaussiangrasper_amd import ops as P
    dev = torch.device("cuda:0")
    viewdirs, coeffs, depths, normals = _inputs(n, k, seed=5)
    tail_ref, mask_ref = oracle.shade_tail_fwd(deg, viewdirs, coeffs, depths, normals)
    t = lambda a: torch.from_numpy(a).to(dev)
    sh, d, nr = t(coeffs).requires_grad_(True), t(depths).requires_grad_(True), t(normals).requires_grad_(True)
    tail = P.ShadeTail.apply(deg, t(viewdirs), sh, d, nr)
    np.testing.assert_array_equal(tail.detach().cpu().numpy(), tail_ref)
    rec = np.random.default_rng(9).standard_normal((n, 13)).astype(np.float32)
    rec_t = t(rec)
    tail.backward(rec_t[:, 6:])                       # a strided view: read in place
    wide = np.zeros((n, 13), np.float32)
    wide[:, :7] = rec[:, 6:]
    v_sh, v_d, v_n = oracle.shade_tail_bwd(deg, k, viewdirs, wide, mask_ref)
    np.testing.assert_array_equal(sh.grad.cpu().numpy(), v_sh)
    np.testing.assert_array_equal(d.grad.cpu().numpy(), v_d)
    np.testing.assert_array_equal(nr.grad.cpu().numpy(), v_n)
    # gradient sink: the SH gradient is added into the registered buffer
    sh2 = t(coeffs).requires_grad_(True)
    prior = np.random.default_rng(10).standard_normal(coeffs.shape).astype(np.float32)
    buf = t(prior).clone()
    P.clear_grad_sinks()
    P.register_grad_sink(sh2, buf)
    try:
        P.ShadeTail.apply(deg, t(viewdirs), sh2, t(depths), t(normals)).backward(rec_t[:, 6:])
    finally:
        P.clear_grad_sinks()
    assert sh2.grad is None
    np.testing.assert_array_equal(buf.cpu().numpy(), prior + v_sh)


@pytest.mark.gpu
@pytest.mark.parametrize("n,k,deg,views", [(5000, 25, 4, 3), (3000, 16, 3, 2), (70, 25, 4, 20)])
def test_deferred_sh_gradient_equals_view_by_view_accumulation(oracle, n, k, deg, views):
    """register_grad_sink(..., defer=...): ShadeTail keeps each view's SH gradient as its factors and expands all
    kept views in one pass (gg_sh_bwd_multi) at the step's last view or at flush_grad_sinks() — bit-identical to
    adding view after view (gg_shade_tail_bwd accumulate), also into a buffer that already holds something; the
    depth / normal cotangents are the same either way; 20 views exercise the 16-view chunks."""
    from gaussiangrasper_amd import ops as P
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(n + views)
    _, coeffs, depths, normals = _inputs(n, k, seed=2)
    dirs = []
    for _ in range(views):
        v = rng.standard_normal((n, 3)).astype(np.float32)
        dirs.append(v / np.linalg.norm(v, axis=1, keepdims=True))
    recs = [torch.from_numpy(rng.standard_normal((n, 13)).astype(np.float32)).to(dev) for _ in range(views)]
    t = lambda a: torch.from_numpy(a).to(dev)

    def run(mode):
        sh = t(coeffs).requires_grad_(True)
        buf = torch.full((n, k, 3), 0.375, device=dev) if n == 3000 else torch.zeros(n, k, 3, device=dev)
        prior = buf.clone()
        fired = []
        state = {"more": True}
        P.clear_grad_sinks()
        if mode == "immediate":
            P.register_grad_sink(sh, buf, lambda p: fired.append(1))
        else:
            P.register_grad_sink(sh, buf, lambda p: fired.append(1), defer=lambda: state["more"])
        side = []
        try:
            for v in range(views):
                d, nr = t(depths).requires_grad_(True), t(normals).requires_grad_(True)
                if mode == "last-view":
                    state["more"] = v < views - 1
                P.ShadeTail.apply(deg, t(dirs[v]), sh, d, nr).backward(recs[v][:, 6:])
                side.append((d.grad.clone(), nr.grad.clone()))
            if mode == "flush":
                assert len(fired) == 0 and torch.equal(buf, prior)            # nothing expanded yet
                P.flush_grad_sinks()
        finally:
            P.clear_grad_sinks()
        assert sh.grad is None
        return buf.cpu().numpy(), fired, side

    ref, fired_ref, side_ref = run("immediate")
    assert len(fired_ref) == views
    for mode in ("last-view", "flush"):
        got, fired, side = run(mode)
        np.testing.assert_array_equal(got, ref)
        assert len(fired) == 1
        for (d0, n0), (d1, n1) in zip(side_ref, side):
            assert torch.equal(d0, d1) and torch.equal(n0, n1)


@pytest.mark.gpu
def test_deferred_sh_gradient_through_the_shim_routes_operator(oracle):
    """the same deferral in ops.SphericalHarmonics (the shim route: the caller's clamp sits behind the operator, its
    backward hands over the already masked colour cotangent)"""
    from gaussiangrasper_amd import ops as P
    dev = torch.device("cuda:0")
    n, k, deg, views = 4000, 25, 4, 4
    rng = np.random.default_rng(1)
    _, coeffs, _, _ = _inputs(n, k, seed=6)
    t = lambda a: torch.from_numpy(a).to(dev)
    dirs = [rng.standard_normal((n, 3)).astype(np.float32) for _ in range(views)]
    cots = [t(rng.standard_normal((n, 3)).astype(np.float32)) for _ in range(views)]

    def run(deferred):
        sh = t(coeffs).requires_grad_(True)
        buf = torch.zeros(n, k, 3, device=dev)
        state = {"more": True}
        P.clear_grad_sinks()
        P.register_grad_sink(sh, buf, None, defer=(lambda: state["more"]) if deferred else None)
        try:
            for v in range(views):
                state["more"] = v < views - 1
                rgb = torch.clamp(P.SphericalHarmonics.apply(deg, t(dirs[v]), sh) + 0.5, 0.0, 1.0)
                rgb.backward(cots[v])
        finally:
            P.clear_grad_sinks()
        return buf.cpu().numpy()

    np.testing.assert_array_equal(run(True), run(False))
