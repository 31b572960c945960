"""`get_loss_dict` / `fea_up` of the plugin's model class: with `fused_training` the subclass
`train.sh` loads computes the reference's loss dictionary (nerfstudio/models/gaussian_splatting.py:841-935) on the fused
loss kernels and swaps `fea_up` (:258) for the MLP kernels' module — behind the UNCHANGED caller
(pipelines/base_pipeline.py:313-329 calls model.get_loss_dict(outputs, batch, metrics_dict)).

Both tests hold the override to `_reference_loss_dict` below: the reference's lines :846-935 written out here (same
statements, same order of random draws), with the reference's SSIM (`pytorch_msssim`, absent) replaced by its published
algorithm (tests/test_image_loss.py).  CPU: oracle-backed rasterizer operators + a torch loss namespace — the host logic
(ground-truth preparation at another resolution, masks, sampling, keys, regularisers, the downscale branch).  GPU: the
product operators and gaussiangrasper_amd.losses, values and gradients of all parameters."""
import sys
import types

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from gaussiangrasper_amd.camera import ring_cameras
from gaussiangrasper_amd.scene import make_scene
from gaussiangrasper_amd.stub import (StubCameras, StubGaussianSplattingModel, default_config, sampling_in_mask,
                                      sampling_pairs_in_mask)
from test_image_loss import _reference_main_loss_dev

KEYS = ("main_loss", "feature_loss", "up_loss", "depth_loss", "normal_loss", "sh_reg", "scale_reg")
PARAMS = ("means", "scales", "quats", "opacities", "colors_all", "feature")


def _cos(e1, e2):      # gaussian_splatting.py:113-118
    return 1 - (F.normalize(e1, dim=0) * F.normalize(e2, dim=0)).sum(dim=0).mean()


def torch_loss_ops():
    """the four entry points of gaussiangrasper_amd.losses as plain torch (any device): what the CPU test plugs in"""
    def main_loss(rgb, gt_img, valid_mask=None, ssim_lambda=0.2):
        return _reference_main_loss_dev(rgb, gt_img, valid_mask, ssim_lambda)

    def depth_normal_loss(depth, gt_depth, normal, gt_normal, depth_mask=None):
        m = depth_mask
        n, d = normal.permute(2, 0, 1), depth.permute(2, 0, 1)
        normal_loss = 0.5 * F.mse_loss(n[:, m], gt_normal[:, m], reduction="mean") + 0.5 * _cos(n[:, m], gt_normal[:, m])
        return F.l1_loss(d[m[None]], gt_depth.reshape(1, *m.shape)[m[None]], reduction="mean"), normal_loss

    def gather_pixels(image, *sets):
        return tuple(image[p[:, 0], p[:, 1]] for p in sets)
    return types.SimpleNamespace(main_loss=main_loss, depth_normal_loss=depth_normal_loss,
                                 cosine_similarity_loss=_cos, gather_pixels=gather_pixels)


def _reference_loss_dict(model, outputs, batch):
    """gaussian_splatting.py:846-935, statement by statement, on clones where the reference writes in place"""
    dev = model.device
    d = model._get_downscale_factor()
    if d > 1:
        newsize = [batch["image"].shape[0] // d, batch["image"].shape[1] // d]
        gt_img = F.interpolate(batch["image"].permute(2, 0, 1)[None], size=tuple(newsize), mode="bilinear",
                               align_corners=False, antialias=False)[0].permute(1, 2, 0)      # TF.resize(antialias=None)
    else:
        gt_img = batch["image"]
    gt_img = gt_img.to(dev).clone()
    size = (gt_img.shape[0], gt_img.shape[1])
    gt_normal = batch["normal"].permute(2, 0, 1).unsqueeze(0).to(dev)
    gt_normal = F.interpolate(gt_normal, size=size, mode='bilinear').squeeze(0)
    gt_normal = F.normalize(gt_normal, dim=0)
    gt_depth = batch["depth"].permute(2, 0, 1).unsqueeze(0).to(dev)
    depth_mask = (gt_depth > 0.05) * 1.0
    gt_depth = F.interpolate(gt_depth, size=size, mode='bilinear').squeeze(0)
    depth_mask = F.interpolate(depth_mask, size=size, mode='nearest').squeeze(0)
    gt_mask = batch["sam_mask"].to(dev)
    gt_mask = F.interpolate(gt_mask.float().unsqueeze(0).unsqueeze(0), size=size, mode='nearest').squeeze(0).squeeze(0)
    valid_mask = batch["valid_mask"].to(dev)
    valid_mask = F.interpolate(valid_mask.float().unsqueeze(0).unsqueeze(0), size=size,
                               mode='nearest').squeeze(0).squeeze(0)
    depth_mask = depth_mask * valid_mask
    depth_mask = depth_mask > 0
    valid_mask = valid_mask > 0
    gt_mask[~valid_mask] = -1.0
    gt_fea = batch["feature"].permute(2, 0, 1).float().to(dev)
    gt_fea = F.interpolate(gt_fea.unsqueeze(0), size=size, mode='nearest').squeeze(0)
    normal = outputs["normal"].permute(2, 0, 1)
    depth = outputs["depth"].permute(2, 0, 1)
    normal_loss = 0.5 * F.mse_loss(normal[:, depth_mask[0]], gt_normal[:, depth_mask[0]], reduction='mean') + \
        0.5 * _cos(normal[:, depth_mask[0]], gt_normal[:, depth_mask[0]])
    depth_loss = F.l1_loss(depth[depth_mask], gt_depth[depth_mask], reduction='mean')
    main_loss = _reference_main_loss_dev(outputs["rgb"], gt_img, valid_mask, model.config.ssim_lambda)[0]   # :882-885, :931
    feature = outputs["feature"]
    selected_pairs = sampling_pairs_in_mask(gt_mask, 800)
    selected_points = sampling_in_mask(gt_mask, 1000)
    fea_loss = 0
    for i in range(len(selected_pairs)):
        f1 = feature[selected_pairs[i][0][:, 0], selected_pairs[i][0][:, 1]]
        f2 = feature[selected_pairs[i][1][:, 0], selected_pairs[i][1][:, 1]]
        fea_loss += _cos(f1.permute(1, 0), f2.permute(1, 0))
    fea_loss = fea_loss / len(selected_pairs)
    fea_up = model.fea_up(feature[selected_points[:, 0], selected_points[:, 1], :]).permute(1, 0)
    up_loss = _cos(fea_up, gt_fea[:, selected_points[:, 0], selected_points[:, 1]])
    if model.step % 10 == 0:
        sh_reg = model.colors_all[:, 1:, :].norm(dim=1).mean()
        scale_exp = torch.exp(model.scales)
        scale_reg = torch.maximum(scale_exp.amax(dim=-1) / scale_exp.amin(dim=-1),
                                  torch.tensor(model.config.max_gauss_ratio)) - model.config.max_gauss_ratio
        scale_reg = 0.1 * scale_reg.mean()
    else:
        sh_reg = torch.tensor(0.0).to(dev)
        scale_reg = torch.tensor(0.0).to(dev)
    return {"main_loss": main_loss, "feature_loss": fea_loss, "up_loss": up_loss, "depth_loss": depth_loss,
            "normal_loss": normal_loss, "sh_reg": sh_reg, "scale_reg": scale_reg}


def _batch(h, w, seed, device="cpu", gt_scale=1.0):
    """what datasets/base_dataset.py:92-124 hands over: image (H, W, 3), normal (h', w', 3), depth (h', w', 1), sam_mask
    (h', w') of labels, valid_mask (h', w'), feature (h'', w'', 512) — the side inputs at their OWN resolutions"""
    g = torch.Generator().manual_seed(seed)
    hs, ws = int(h * gt_scale), int(w * gt_scale)
    yy, xx = torch.meshgrid(torch.arange(hs), torch.arange(ws), indexing="ij")
    sam = ((yy * 3) // hs + 3 * ((xx * 2) // ws)).float() - 1.0            # labels -1 .. 4 in blocks
    valid = torch.rand(hs, ws, generator=g) > 0.1
    depth = torch.rand(hs, ws, 1, generator=g) * 4 + 0.2
    depth[torch.rand(hs, ws, generator=g) > 0.9] = 0.0                     # holes in the sensor depth
    b = {"image": torch.rand(h, w, 3, generator=g), "normal": torch.randn(hs, ws, 3, generator=g), "depth": depth,
         "sam_mask": sam, "valid_mask": valid, "feature": torch.randn(max(hs // 4, 2), max(ws // 4, 2), 512, generator=g)}
    return {k: v.to(device) for k, v in b.items()}


def _model(ops, loss_ops, mlp_class, n, h, w, device, step, feature_dim=32):
    from gaussiangrasper_amd.plugin import make_fused_model_class
    sc = make_scene(n, feature_dim=feature_dim, config_index=9)
    sc.scales.add_(1.6)
    Model = make_fused_model_class(StubGaussianSplattingModel, ops=ops, fused_training=True, loss_ops=loss_ops,
                                   mlp_class=mlp_class)
    torch.manual_seed(5)
    m = Model(sc, config=default_config(), step=step).to(device).train()
    cam = StubCameras.from_view(ring_cameras(4, h, w)[1], device=device)
    return m, cam


def _compare(m, cam, batch, rtol, gtol):
    out = m.get_outputs(cam)
    torch.manual_seed(11)
    got = m.get_loss_dict(out, batch)
    assert tuple(got) == KEYS
    sum(got.values()).backward()
    g_got = {k: getattr(m, k).grad.clone() for k in PARAMS}
    g_got.update({"fea_up." + k: p.grad.clone() for k, p in m.fea_up.named_parameters()})
    m.zero_grad(set_to_none=True)
    out = m.get_outputs(cam)
    torch.manual_seed(11)
    want = _reference_loss_dict(m, out, batch)
    sum(want.values()).backward()
    for k in KEYS:
        a, b = float(got[k].detach()), float(want[k].detach())
        assert abs(a - b) <= rtol * max(1.0, abs(b)), (k, a, b)
    for k, g in g_got.items():
        ref = (m.fea_up.get_parameter(k[7:]) if k.startswith("fea_up.") else getattr(m, k)).grad
        scale = float(ref.abs().max())
        assert scale > 0, k
        assert float((g - ref).abs().max()) <= gtol * scale, (k, float((g - ref).abs().max()), scale)
    return got


@pytest.mark.parametrize("step,gt_scale", [(30000, 1.0), (30001, 0.5), (0, 0.75)])
def test_plugin_get_loss_dict_host_logic_on_cpu(step, gt_scale):
    """oracle-backed operators, torch losses: keys, ground-truth preparation at another resolution, masks, the order of
    the random draws, the regularisers (step % 10), the training downscale branch (step 0: d = 4)"""
    import oracle_ops
    h, w = 64, 96
    m, cam = _model(oracle_ops, torch_loss_ops(), None, 500, h, w, "cpu", step, feature_dim=8)
    assert type(m.fea_up).__module__ == "gaussiangrasper_amd.stub"       # mlp_class None keeps the reference's module
    d = m._get_downscale_factor()
    got = _compare(m, cam, _batch(h, w, 3, gt_scale=gt_scale), 1e-6, 1e-5)
    assert (float(got["sh_reg"].detach()) != 0.0) == (step % 10 == 0)
    assert d == (4 if step == 0 else 1)
    # GG_FUSED_TRAINING=0 / fused_training False: the reference's own method runs (the stub has none: AttributeError)
    from gaussiangrasper_amd.plugin import make_fused_model_class
    plain = make_fused_model_class(StubGaussianSplattingModel, ops=oracle_ops)(make_scene(50, feature_dim=8))
    with pytest.raises(AttributeError):
        plain.get_loss_dict({}, {})


@pytest.mark.gpu
@pytest.mark.parametrize("h,w,step", [(96, 128, 30000), (120, 160, 30007)])
def test_plugin_get_loss_dict_on_the_gpu(h, w, step):
    """the product: HIP rasterizer, fused loss kernels, fea_up on the MLP kernels — every loss and the gradient of every
    parameter (the six Gaussian groups and fea_up's) against the reference's lines on the same device"""
    from gaussiangrasper_amd import mlp, ops as P
    dev = torch.device("cuda:0")
    m, cam = _model(P, None, "default", 4000, h, w, dev, step)
    assert isinstance(m.fea_up, mlp.MLP) and set(m.fea_up.state_dict()) == {
        "layers.0.weight", "layers.0.bias", "layers.2.weight", "layers.2.bias"}
    P.clear_bin_cache()
    _compare(m, cam, _batch(h, w, 7, device=dev, gt_scale=0.5), 2e-5, 3e-4)


def test_fea_up_swap_keeps_the_reference_state_dict():
    """a checkpoint written with the reference's module loads into the swapped one and back (trainer.py:428-456 saves
    `_model.fea_up.layers.{0,2}.{weight,bias}`)"""
    from gaussiangrasper_amd import mlp
    from gaussiangrasper_amd.plugin import make_fused_model_class
    from gaussiangrasper_amd.stub import MLP as RefMLP
    sc = make_scene(20, feature_dim=32, config_index=9)
    torch.manual_seed(2)
    ref = StubGaussianSplattingModel(sc)
    fused = make_fused_model_class(StubGaussianSplattingModel, fused_training=True)(sc)
    assert isinstance(ref.fea_up, RefMLP) and isinstance(fused.fea_up, mlp.MLP)
    sd = {k: v for k, v in ref.state_dict().items() if k.startswith("fea_up.")}
    assert set(sd) == {k for k in fused.state_dict() if k.startswith("fea_up.")}
    fused.load_state_dict(ref.state_dict())
    for k, v in sd.items():
        assert torch.equal(fused.state_dict()[k], v)


def test_device_sampling_draws_the_reference_law():
    """gaussiangrasper_amd.sampling (the plugin's opt-in replacement for the reference's host-randperm helpers): same
    shapes and counts as the helpers of :120-148 on the same mask, every sample inside its label, distinct within a
    draw, the two members of a pair independent draws, all pixels of a label reachable"""
    from gaussiangrasper_amd import sampling as S
    g = torch.Generator().manual_seed(2)
    mask = torch.randint(-1, 4, (37, 53), generator=g).float()
    mask[:3] = 7.0                                              # a label value that is not consecutive
    mask[30:, 40:] = -1.0
    torch.manual_seed(1)
    ref_pts, ref_pairs = sampling_in_mask(mask, 100), sampling_pairs_in_mask(mask, 60)
    pts, pairs = S.sampling_in_mask(mask, 100), S.sampling_pairs_in_mask(mask, 60)
    assert pts.shape == ref_pts.shape and pts.dtype == torch.int64
    assert len(pairs) == len(ref_pairs) == 5
    labels = [v for v in torch.unique(mask).tolist() if v > -1]
    per = 100 // len(torch.unique(mask).tolist()[1:])
    assert per == 100 // (len(torch.unique(mask)) - 1)
    off = 0
    for lab, (a, b), (ra, rb) in zip(labels, pairs, ref_pairs):
        assert a.shape == ra.shape and b.shape == rb.shape
        for s in (a, b, pts[off:off + min(per, int((mask == lab).sum()))]):
            assert bool((mask[s[:, 0], s[:, 1]] == lab).all())
            assert len({(int(r), int(c)) for r, c in s.tolist()}) == s.shape[0]      # without replacement
        assert not torch.equal(a, b)
        off += min(per, int((mask == lab).sum()))
    assert off == pts.shape[0]
    # a label with fewer pixels than asked for: all of them, each once
    small = torch.full((6, 6), -1.0)
    small[1, 2] = small[4, 5] = small[0, 0] = 0.0
    small[3, 3] = 1.0
    s = S.sampling_in_mask(small, 1000)
    assert sorted(map(tuple, s.tolist())) == [(0, 0), (1, 2), (3, 3), (4, 5)]
    hits = torch.zeros(6, 6)
    for _ in range(40):
        p = S.sampling_pairs_in_mask(small, 1)[0][0]
        hits[p[0, 0], p[0, 1]] += 1
    assert bool((hits[small == 0] > 0).all()) and float(hits[small != 0].sum()) == 0


def test_plugin_get_loss_dict_with_device_sampling_on_cpu():
    """device_sampling=True: same keys, finite losses, gradients everywhere the reference's sampling gives them"""
    import oracle_ops
    from gaussiangrasper_amd.plugin import make_fused_model_class
    h, w = 64, 96
    sc = make_scene(500, feature_dim=8, config_index=9)
    sc.scales.add_(1.6)
    Model = make_fused_model_class(StubGaussianSplattingModel, ops=oracle_ops, fused_training=True,
                                   loss_ops=torch_loss_ops(), mlp_class=None, device_sampling=True)
    torch.manual_seed(5)
    m = Model(sc, config=default_config(), step=30000).train()
    cam = StubCameras.from_view(ring_cameras(4, h, w)[1])
    state = torch.random.get_rng_state()
    got = m.get_loss_dict(m.get_outputs(cam), _batch(h, w, 3))
    assert tuple(got) == KEYS and all(bool(torch.isfinite(v.detach()).all()) for v in got.values())
    sum(got.values()).backward()
    assert all(getattr(m, k).grad is not None and float(getattr(m, k).grad.abs().max()) > 0 for k in PARAMS)
    assert float(got["feature_loss"].detach()) != 0.0 and float(got["up_loss"].detach()) != 0.0
    del state
