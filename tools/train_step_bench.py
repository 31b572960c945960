#!/usr/bin/env python3
"""One training ITERATION end to end on one MI355X, every piece of SURVEY §8 in the loop (a measurement tool; the
headline metric stays bench.py's operator path with synthetic cotangents):

    per view (the reference trains one view per step; here `--views` views share a step as in bench.py):
        outputs = model(camera)                 the plugin's model class (get_outputs: ActivateGaussians, ProjectGaussians,
                                                ShadeTail, RasterizeSegments)
        loss_dict = model.get_loss_dict(outputs, batch)      (r04: the class's own method, as pipelines/base_pipeline.py:
                                                325-326 calls it — ground-truth preparation, masks, sampling and all)
            main_loss   = (1 - 0.2) L1 + 0.2 (1 - SSIM)          gaussian_splatting.py:882-885, :931
            depth_loss  = L1, normal_loss = 0.5 mse + 0.5 cosine   :879-880   over the masked pixels
            feature_loss: cosine similarity of 800 sampled pixel pairs per mask, up_loss: fea_up MLP on 1000 sampled
                          pixels against the 512-dim target (:905-918); sh_reg / scale_reg (:920-929)
        sum(loss_dict.values()).backward()      (trainer.py:474-475)
    one Adam step over the six parameter groups (engine/optimizers.py:158-171; lrs method_configs.py:618-660)

and the same iteration with the caller's torch code for the pieces that have fused replacements (the same
get_loss_dict with a torch loss namespace: SSIM main loss through grouped conv2d, boolean-index depth / normal losses,
torch cosine losses, nn.Sequential fea_up autograd; torch.optim.Adam), on the same rasterizer.
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

from gaussiangrasper_amd import ops  # noqa: E402
from gaussiangrasper_amd.camera import ring_cameras  # noqa: E402
from gaussiangrasper_amd.dist import GradBucket  # noqa: E402
from gaussiangrasper_amd.optim import FusedAdam, fused_step  # noqa: E402
from gaussiangrasper_amd.scene import make_scene  # noqa: E402
from test_plugin_losses import torch_loss_ops  # noqa: E402

LRS = dict(means=1.6e-4, scales=0.005, quats=0.001, opacities=0.05, colors_all=5e-4, feature=5e-4)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--views", type=int, default=8)
    ap.add_argument("--skip-torch", action="store_true", help="only the fused variant (clean kernel profiles)")
    ap.add_argument("--lib", default=None, help="A/B: another build of the library")
    ap.add_argument("--gc", default="default", choices=["default", "freeze", "off"],
                    help="Python's cyclic collector during the timed steps: as it is, after gc.freeze() (everything "
                         "alive after the warm-up moved out of the collector's way), or disabled")
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
    # render AND losses go through the class train.sh loads (plugin.FusedGaussianSplattingModel.get_outputs /
    # .get_loss_dict, on stub.py's stand-ins for nerfstudio's base model and Cameras); two instances over the SAME six
    # Parameters: the fused one (gaussiangrasper_amd.losses, fea_up on the MLP kernels) and one whose loss namespace and
    # fea_up are plain torch (the reference's expressions)
    from gaussiangrasper_amd.plugin import make_fused_model_class
    from gaussiangrasper_amd.stub import StubCameras, StubGaussianSplattingModel
    torch.manual_seed(3)
    # three instances over the SAME six Parameters:
    #   model    fused losses, fea_up on the MLP kernels, pixel samples from gaussiangrasper_amd.sampling (device generator)
    #   model_r  the same with the reference module's sampling helpers (nine host torch.randperm per view: the default,
    #            same draws as the reference)
    #   model_t  loss namespace and fea_up plain torch, the reference's sampling helpers: the reference's own expressions
    mk = lambda **kw: make_fused_model_class(StubGaussianSplattingModel, fused_training=True, **kw)(scene).to(dev).train()
    model = mk(device_sampling=True)
    model_r = mk(device_sampling=False)
    model_t = mk(loss_ops=torch_loss_ops(), mlp_class=None, device_sampling=False)
    names = ("means", "scales", "quats", "opacities", "colors_all", "feature")
    for n_ in names:
        setattr(scene, n_, getattr(model, n_))
        setattr(model_t, n_, getattr(model, n_))
        setattr(model_r, n_, getattr(model, n_))
    model_t.fea_up.load_state_dict(model.fea_up.state_dict())
    model_r.fea_up = model.fea_up
    cams = [StubCameras.from_view(v, device=dev, cam_idx=i) for i, v in enumerate(views)]
    g = torch.Generator(device="cpu").manual_seed(7)
    # a synthetic batch of the shapes datasets/base_dataset.py:92-124 hands over, resident in HBM: side inputs at half
    # the image resolution, the CLIP feature map at an eighth
    hs, ws = h // 2, w // 2
    yy, xx = torch.meshgrid(torch.arange(hs), torch.arange(ws), indexing="ij")
    depth = torch.rand(hs, ws, 1, generator=g) * 5 + 0.5
    depth[torch.rand(hs, ws, generator=g) > 0.95] = 0.0
    batch = {"image": torch.rand(h, w, 3, generator=g), "normal": torch.randn(hs, ws, 3, generator=g), "depth": depth,
             "sam_mask": ((yy * 2) // hs + 2 * ((xx * 2) // ws)).float() - 1.0,         # labels -1, 0, 1, 2
             "valid_mask": torch.rand(hs, ws, generator=g) > 0.05,
             "feature": torch.randn(h // 8, w // 8, 512, generator=g)}
    batch = {k: v.to(dev) for k, v in batch.items()}

    bucket = GradBucket(scene.params())
    bucket.enable_direct(ops, defer_sh=True)     # as bench.py: the SH gradient of the step's views expanded once

    def optimizers(cls):
        return [cls([getattr(scene, n)], lr=LRS[n], eps=1e-15) for n in names] + \
               [cls(list((model if cls is FusedAdam else model_t).fea_up.parameters()), lr=1e-3, eps=1e-15)]

    def iteration(fused, opts):
        m = {True: model, "reference-sampling": model_r, False: model_t}[fused]
        bucket.zero_()
        opts[-1].zero_grad(set_to_none=True)        # fea_up's parameters (the Gaussians' gradients live in the bucket)
        for k in range(len(views)):
            if k == len(views) - 1:
                bucket.arm()                          # the step's last backward
            out = m(cams[k])
            loss_dict = m.get_loss_dict(out, batch)
            sum(loss_dict.values()).backward()
        bucket.finish()
        if fused is not False:
            fused_step(opts)
        else:
            for o in opts:
                o.step()

    def timed(fused, opts):
        import gc
        for _ in range(a.warmup):
            iteration(fused, opts)
        torch.cuda.synchronize()
        gc.enable()
        gc.unfreeze()
        if a.gc == "freeze":
            gc.collect()
            gc.freeze()
        elif a.gc == "off":
            gc.disable()
        t0 = time.perf_counter()
        for _ in range(a.steps):
            iteration(fused, opts)
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / a.steps

    t_fused = timed(True, optimizers(FusedAdam))
    t_fused_r = timed("reference-sampling", optimizers(FusedAdam)) if not a.skip_torch else float("nan")
    t_torch = timed(False, optimizers(torch.optim.Adam)) if not a.skip_torch else float("nan")
    print(json.dumps({
        "workload": "%d Gaussians, %dx%d, %d views per optimizer step: model(camera) + model.get_loss_dict(outputs, batch) "
                    "of the plugin's class (main / depth / normal / feature / up losses + regularisers) + backward, one "
                    "Adam step over 6 Gaussian groups + fea_up" % (a.points, w, h, a.views),
        "steps": a.steps, "views_per_step": a.views, "python_gc": a.gc,
        "fused_losses_and_adam": {"ms_per_step": round(1e3 * t_fused, 2), "views_per_s": round(a.views / t_fused, 1),
                                  "sampling": "gaussiangrasper_amd.sampling (device generator; GG_DEVICE_SAMPLING=1)"},
        "fused_losses_and_adam_reference_sampling": {
            "ms_per_step": round(1e3 * t_fused_r, 2), "views_per_s": round(a.views / t_fused_r, 1),
            "sampling": "the reference's helpers (gaussian_splatting.py:120-148: nine host torch.randperm over a label's "
                        "pixel count per view; the plugin's default — same draws as the reference)"},
        "torch_losses_and_adam": {"ms_per_step": round(1e3 * t_torch, 2), "views_per_s": round(a.views / t_torch, 1)},
        "note": "same HIP rasterizer and the same get_loss_dict in both; 'torch' = its loss namespace is plain torch (SSIM "
                "main loss through grouped conv2d, boolean-index depth / normal losses, torch cosine losses, nn.Sequential "
                "fea_up autograd), its pixel samples come from the reference's helpers and the optimizers are seven "
                "torch.optim.Adam; bench.py's headline excludes the losses",
    }))


if __name__ == "__main__":
    main()
