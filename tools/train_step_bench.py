#!/usr/bin/env python3
"""One training ITERATION end to end on one MI355X, every piece of SURVEY §8 in the loop (a measurement tool; the
headline metric stays bench.py's operator path with synthetic cotangents):

    per view (the reference trains one view per step; here `--views` views share a step as in bench.py):
        render through the plugin's model class (get_outputs: ActivateGaussians, ProjectGaussians, ShadeTail,
        RasterizeSegments)
        main_loss   = (1 - 0.2) L1 + 0.2 (1 - SSIM)          gaussian_splatting.py:882-885, :931
        depth_loss  = L1, normal_loss = 0.5 mse + 0.5 cosine   :879-880   over the masked pixels
        feature_loss: cosine similarity of 800 sampled pixel pairs, up_loss: fea_up MLP on 1000 sampled pixels
                      against a 512-dim target (:905-918)
        backward
    one Adam step over the six parameter groups (engine/optimizers.py:158-171; lrs method_configs.py:618-660)

and the same iteration with the caller's torch code for the pieces that have fused replacements (SSIM main loss
through grouped conv2d, boolean-index depth / normal losses, torch cosine / MLP autograd, torch.optim.Adam), on
the same rasterizer.
Prints one JSON object.  Usage: python tools/train_step_bench.py [--steps 3] [--views 8] [--points 1000000]"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "shim"), os.path.join(ROOT, "tests")]
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402

from gaussiangrasper_amd import losses, ops  # noqa: E402
from gaussiangrasper_amd.camera import ring_cameras  # noqa: E402
from gaussiangrasper_amd.dist import GradBucket  # noqa: E402
from gaussiangrasper_amd.mlp import MLP  # noqa: E402
from gaussiangrasper_amd.optim import FusedAdam, fused_step  # noqa: E402
from gaussiangrasper_amd.pipeline import render_view  # noqa: E402
from gaussiangrasper_amd.scene import make_scene  # noqa: E402
from test_image_loss import _reference_main_loss_dev  # noqa: E402

LRS = dict(means=1.6e-4, scales=0.005, quats=0.001, opacities=0.05, colors_all=5e-4, feature=5e-4)


def torch_cosine_loss(e1, e2):      # reference :113-118, embeddings (C, M)
    return 1 - (F.normalize(e1, dim=0) * F.normalize(e2, dim=0)).sum(dim=0).mean()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--views", type=int, default=8)
    ap.add_argument("--skip-torch", action="store_true", help="only the fused variant (clean kernel profiles)")
    ap.add_argument("--lib", default=None, help="A/B: another build of the library")
    ap.add_argument("--points", type=int, default=1_000_000)
    ap.add_argument("--height", type=int, default=1200)
    ap.add_argument("--width", type=int, default=1600)
    a = ap.parse_args()
    if a.lib:
        from gaussiangrasper_amd import _lib
        _lib.LIB_PATH = os.path.abspath(a.lib)
        _lib.load(build_if_missing=False)
    dev = torch.device("cuda:0")
    h, w = a.height, a.width
    scene = make_scene(a.points, config_index=3).to(dev)
    views = ring_cameras(a.views, h, w, device=dev)
    # the render goes through the class train.sh loads (plugin.FusedGaussianSplattingModel.get_outputs, on stub.py's
    # stand-ins for nerfstudio's base model and Cameras); its six Parameters are the scene's leaves
    from gaussiangrasper_amd.plugin import make_fused_model_class
    from gaussiangrasper_amd.stub import StubCameras, StubGaussianSplattingModel
    model = make_fused_model_class(StubGaussianSplattingModel, fused_training=True)(scene).train()
    for n_ in ("means", "scales", "quats", "opacities", "colors_all", "feature"):
        setattr(scene, n_, getattr(model, n_))
    cams = [StubCameras.from_view(v, device=dev, cam_idx=i) for i, v in enumerate(views)]
    g = torch.Generator(device="cpu").manual_seed(7)
    # synthetic supervision of the right shapes (resident in HBM)
    gt_rgb = torch.rand(h, w, 3, generator=g).to(dev)
    gt_depth = (torch.rand(h, w, 1, generator=g) * 5 + 0.5).to(dev)
    gt_normal_chw = F.normalize(torch.randn(3, h, w, generator=g), dim=0).to(dev)     # (3, H, W) as the reference
    valid = (torch.rand(h, w, generator=g) > 0.05).to(dev)
    depth_mask = valid & (gt_depth[..., 0] > 0.05)
    pairs = [torch.stack([torch.randint(0, h, (800,), generator=g), torch.randint(0, w, (800,), generator=g)], 1).to(dev)
             for _ in range(2)]
    pts = torch.stack([torch.randint(0, h, (1000,), generator=g), torch.randint(0, w, (1000,), generator=g)], 1).to(dev)
    gt_fea = torch.randn(512, 1000, generator=g).to(dev)
    torch.manual_seed(3)
    fea_up = MLP(32, 512, hidden_list=[128]).to(dev)
    fea_up_torch = torch.nn.Sequential(torch.nn.Linear(32, 128), torch.nn.ReLU(), torch.nn.Linear(128, 512)).to(dev)
    fea_up_torch.load_state_dict({k.replace("layers.", ""): v for k, v in fea_up.state_dict().items()})

    bucket = GradBucket(scene.params())
    bucket.enable_direct(ops, defer_sh=True)     # as bench.py: the SH gradient of the step's views expanded once
    names = ("means", "scales", "quats", "opacities", "colors_all", "feature")

    def optimizers(cls):
        return [cls([getattr(scene, n)], lr=LRS[n], eps=1e-15) for n in names] + \
               [cls(list((fea_up if cls is FusedAdam else fea_up_torch).parameters()), lr=1e-3, eps=1e-15)]

    def iteration(fused: bool, opts):
        bucket.zero_()
        opts[-1].zero_grad(set_to_none=True)        # fea_up's parameters (the Gaussians' gradients live in the bucket)
        for k, v in enumerate(views):
            if k == len(views) - 1:
                bucket.arm()                          # the step's last backward
            out = model(cams[k])
            rgb, depth, normal, feature = out["rgb"], out["depth"], out["normal"], out["feature"]
            if fused:
                main_l = losses.main_loss(rgb, gt_rgb, valid, 0.2)[0]
            else:
                main_l = _reference_main_loss_dev(rgb, gt_rgb, valid, 0.2)[0]
            cos = losses.cosine_similarity_loss if fused else torch_cosine_loss
            if fused:
                depth_l, normal_l = losses.depth_normal_loss(depth, gt_depth, normal, gt_normal_chw, depth_mask)
            else:       # the reference's lines :879-880: boolean-index gathers, then elementwise torch
                depth_l = F.l1_loss(depth[depth_mask], gt_depth[depth_mask])
                nrm, gtn = normal.permute(2, 0, 1)[:, depth_mask], gt_normal_chw[:, depth_mask]
                normal_l = 0.5 * F.mse_loss(nrm, gtn) + 0.5 * cos(nrm, gtn)
            if fused:       # one gather (one zero-filled gradient image in the backward) for all sampled sets
                f1, f2, fp = losses.gather_pixels(feature, pairs[0], pairs[1], pts)
            else:           # the reference's three advanced-indexing gathers (:912-917)
                f1 = feature[pairs[0][:, 0], pairs[0][:, 1]]
                f2 = feature[pairs[1][:, 0], pairs[1][:, 1]]
                fp = feature[pts[:, 0], pts[:, 1], :]
            fea_l = cos(f1.permute(1, 0), f2.permute(1, 0))
            up = (fea_up if fused else fea_up_torch)(fp).permute(1, 0)
            up_l = cos(up, gt_fea)
            (main_l + depth_l + normal_l + fea_l + up_l).backward()
        bucket.finish()
        if fused:
            fused_step(opts)
        else:
            for o in opts:
                o.step()

    def timed(fused, opts):
        for _ in range(a.warmup):
            iteration(fused, opts)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(a.steps):
            iteration(fused, opts)
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / a.steps

    t_fused = timed(True, optimizers(FusedAdam))
    t_torch = timed(False, optimizers(torch.optim.Adam)) if not a.skip_torch else float("nan")
    print(json.dumps({
        "workload": "%d Gaussians, %dx%d, %d views per optimizer step: plugin-route render + main / depth / normal / "
                    "feature / up losses + backward, one Adam step over 6 Gaussian groups + fea_up" % (a.points, w, h, a.views),
        "steps": a.steps, "views_per_step": a.views,
        "fused_losses_and_adam": {"ms_per_step": round(1e3 * t_fused, 2), "views_per_s": round(a.views / t_fused, 1)},
        "torch_losses_and_adam": {"ms_per_step": round(1e3 * t_torch, 2), "views_per_s": round(a.views / t_torch, 1)},
        "note": "same HIP rasterizer in both; 'torch' = SSIM main loss through grouped conv2d, boolean-index depth / "
                "normal losses, torch cosine losses, nn.Sequential fea_up autograd, seven torch.optim.Adam; bench.py's "
                "headline excludes the losses",
    }))


if __name__ == "__main__":
    main()
