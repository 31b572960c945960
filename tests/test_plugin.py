"""The nerfstudio plugin route (SURVEY 8f-1 / 8b "second route"): `gaussiangrasper_amd.plugin` builds a
subclass of the reference model whose get_outputs renders the four images from ONE fused rasterize call.
nerfstudio itself is not importable here, so the subclass is built on a stub with the attributes the
reference model has (nerfstudio/models/gaussian_splatting.py:231-313, 599-619) and a stub camera with the
`Cameras` fields get_outputs reads (:655-682); operators are the oracle-backed ones (CPU)."""
import copy
import sys
import types

import pytest
import torch

from gaussiangrasper_amd.camera import ring_cameras
from gaussiangrasper_amd.pipeline import render_view
from gaussiangrasper_amd.scene import make_scene


class _Cam:
    """the fields of nerfstudio.cameras.cameras.Cameras that get_outputs touches"""

    def __init__(self, view, c2w):
        t = lambda v: torch.tensor([[v]])
        self.camera_to_worlds = c2w[None, :3, :]
        self.fx, self.fy, self.cx, self.cy = t(view.fx), t(view.fy), t(view.cx), t(view.cy)
        self.width, self.height = torch.tensor([[view.width]]), torch.tensor([[view.height]])
        self.shape = (1,)
        self.rescales = []

    def rescale_output_resolution(self, f):
        self.rescales.append(f)


class _StubBase(torch.nn.Module):
    """what the subclass uses of GaussianSplattingModel"""

    def __init__(self, sc):
        super().__init__()
        for k in ("means", "scales", "quats", "opacities", "colors_all", "feature"):
            setattr(self, k, torch.nn.Parameter(getattr(sc, k).clone()))
        self.config = types.SimpleNamespace(sh_degree=4, sh_degree_interval=1000)
        self.step = 30000
        self.crop_box = None
        self.back_color = torch.zeros(3)
        self.feature_dim = sc.feature.shape[1]
        self.camera_optimizer = types.SimpleNamespace(apply_to_camera=lambda cam: None)

    @property
    def device(self):
        return self.means.device

    def _get_downscale_factor(self):
        return 1


def _c2w_of(view):
    w2c = view.viewmat.clone()
    R = w2c[:3, :3].T @ torch.diag(torch.tensor([1.0, -1.0, -1.0]))      # undo the pi rotation about x
    c2w = torch.eye(4)
    c2w[:3, :3] = R
    c2w[:3, 3] = view.cam_pos
    return c2w


def test_fused_plugin_model_renders_the_same_images_as_the_four_call_sequence():
    import oracle_ops
    from gaussiangrasper_amd.plugin import make_fused_model_class
    sc = make_scene(500, feature_dim=8, config_index=9)
    sc.scales.add_(1.7)
    from gaussiangrasper_amd.camera import view_from_c2w
    v0 = ring_cameras(4, 40, 56)[1]
    c2w = _c2w_of(v0)
    f32 = lambda x: float(torch.tensor(x, dtype=torch.float32))         # Cameras holds fp32 intrinsics
    view = view_from_c2w(c2w, f32(v0.fx), f32(v0.fy), f32(v0.cx), f32(v0.cy), 40, 56)   # same c2w both sides
    want = render_view(copy.deepcopy(sc), view, oracle_ops)
    Model = make_fused_model_class(_StubBase, ops=oracle_ops)
    m = Model(sc)
    m.train()
    cam = _Cam(view, c2w)
    out = m.get_outputs(cam)
    assert set(out) == {"rgb", "feature", "depth", "normal", "normal_vis", "feature_vis"}
    for k in ("rgb", "feature", "depth", "normal"):
        assert out[k].shape == want[k].shape, k
        assert torch.equal(out[k], want[k]), k
    assert out["feature_vis"].shape == (40, 56, 3) and out["normal_vis"].shape == (40, 56, 3)
    assert cam.rescales == [1.0, 1]                                   # resolution rescaled and restored
    assert m.last_size == (40, 56) and m.radii.shape == (500,) and m.normals.shape == (500, 3)
    # side channel of the densification statistics (:376-393): xys keeps its gradient
    (out["rgb"].sum() + out["feature"].sum() + out["depth"].sum() + out["normal"].sum()).backward()
    assert m.xys.grad is not None and m.xys.grad.shape == (500, 2) and float(m.xys.grad.abs().sum()) > 0
    for k in ("means", "scales", "quats", "opacities", "colors_all", "feature"):
        assert getattr(m, k).grad is not None and float(getattr(m, k).grad.abs().sum()) > 0, k
    # eval mode: background override hook and the not-a-camera guard
    m.eval()
    assert m.get_outputs(object()) == {}
    out2 = m.get_outputs(_Cam(view, c2w))
    assert torch.equal(out2["rgb"], out["rgb"])


def test_plugin_specifications_replace_only_the_model_target(monkeypatch):
    """gaussian_splatting() copies the reference's own TrainerConfig for the method and swaps
    pipeline.model._target (registry route: NERFSTUDIO_METHOD_CONFIGS, plugins/registry.py:53-75)"""
    from gaussiangrasper_amd import plugin

    class MethodSpecification:
        def __init__(self, config, description):
            self.config, self.description = config, description

    class GaussianSplattingModel(_StubBase):
        pass

    ref_cfg = types.SimpleNamespace(method_name="gaussian-splatting", max_num_iterations=30000,
                                    pipeline=types.SimpleNamespace(model=types.SimpleNamespace(_target=GaussianSplattingModel)))
    mods = {
        "nerfstudio": types.ModuleType("nerfstudio"),
        "nerfstudio.configs": types.ModuleType("nerfstudio.configs"),
        "nerfstudio.configs.method_configs": types.ModuleType("nerfstudio.configs.method_configs"),
        "nerfstudio.plugins": types.ModuleType("nerfstudio.plugins"),
        "nerfstudio.plugins.types": types.ModuleType("nerfstudio.plugins.types"),
        "nerfstudio.models": types.ModuleType("nerfstudio.models"),
        "nerfstudio.models.gaussian_splatting": types.ModuleType("nerfstudio.models.gaussian_splatting"),
        "nerfstudio.model_components": types.ModuleType("nerfstudio.model_components"),
        "nerfstudio.model_components.renderers": types.ModuleType("nerfstudio.model_components.renderers"),
    }
    mods["nerfstudio.configs.method_configs"].method_configs = {"gaussian-splatting": ref_cfg}
    mods["nerfstudio.plugins.types"].MethodSpecification = MethodSpecification
    mods["nerfstudio.models.gaussian_splatting"].GaussianSplattingModel = GaussianSplattingModel
    mods["nerfstudio.model_components.renderers"].BACKGROUND_COLOR_OVERRIDE = None
    mods["nerfstudio.model_components"].renderers = mods["nerfstudio.model_components.renderers"]
    for k, v in mods.items():
        monkeypatch.setitem(sys.modules, k, v)
    monkeypatch.setattr(plugin, "_model_class", None)
    spec = plugin.gaussian_splatting()
    assert isinstance(spec, MethodSpecification) and spec.config.method_name == "gaussian-splatting"
    assert issubclass(spec.config.pipeline.model._target, GaussianSplattingModel)
    assert spec.config.pipeline.model._target.__name__ == "FusedGaussianSplattingModel"
    assert spec.config.max_num_iterations == 30000
    assert ref_cfg.pipeline.model._target is GaussianSplattingModel      # the reference's entry is untouched
    assert plugin.gaussian_splatting_amd().config.method_name == "gaussian-splatting-amd"
