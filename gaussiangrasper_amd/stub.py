"""Stand-ins for the two nerfstudio classes the plugin's model touches, for machines without nerfstudio.

nerfstudio is not importable in the build container nor on the GPU box (torchvision, tyro, viser ... are
absent), so `bench.py --route plugin` and the tests build
`plugin.make_fused_model_class(StubGaussianSplattingModel)` — the SAME subclass body `train.sh` gets on top of
the real `GaussianSplattingModel` — and drive its `get_outputs` with `StubCameras`.  Only attributes the
reference's own `get_outputs` / callbacks read are restated, with the reference's names:

  StubCameras                 nerfstudio/cameras/cameras.py:69-99 (fields), :935-961 (rescale_output_resolution)
  StubGaussianSplattingModel  nerfstudio/models/gaussian_splatting.py:248-299 (populate_modules: the six Gaussian
                              parameters, `fea_up`, statistics, step, crop_box, back_color, config), :599-603
                              (_get_downscale_factor), :548-571 (get_training_callbacks: after_train, then
                              refinement_after every refine_every steps), :574-586 (step_cb, param groups)
  sampling_in_mask, sampling_pairs_in_mask, MLP
                              the module-level helpers of gaussian_splatting.py that `get_loss_dict` calls (:120-148,
                              :198-213): the plugin's override looks them up in its base class's MODULE, which is the
                              reference's own file under nerfstudio and this one here"""
from __future__ import annotations

import types
from typing import Dict, List, Optional

import torch
from torch.nn import Parameter


def sampling_in_mask(mask, sample_num):
    """Behaviour of gaussian_splatting.py:120-132: for every label > -1 of the mask (ascending, `torch.unique`), up to
    sample_num // (number of labels - 1) of its pixels, drawn by ONE `torch.randperm` over the label's pixel count
    (row-major `torch.where` order) — the same draws from the global RNG, in the same order, as the reference's.
    Returns (M, 2) long (row, col)."""
    mask = mask.detach()
    labels = torch.unique(mask)
    per_label = sample_num // (len(labels) - 1)
    out = []
    for lab in labels:
        if lab > -1:
            rows, cols = torch.where(mask == lab)
            pick = torch.randperm(rows.shape[0])[:min(per_label, rows.shape[0])]
            out.append(torch.stack((rows[pick], cols[pick]), dim=1))
    return torch.cat(out)


def sampling_pairs_in_mask(mask, sample_num):
    """Behaviour of gaussian_splatting.py:134-148: for every label > -1, two independent draws (two `torch.randperm`
    calls, first set first) of up to sample_num of its pixels.  Returns [[first (M, 2), second (M, 2)], ...]."""
    mask = mask.detach()
    pairs = []
    for lab in torch.unique(mask):
        if lab > -1:
            rows, cols = torch.where(mask == lab)
            m = min(sample_num, rows.shape[0])
            one = torch.randperm(rows.shape[0])[:m]
            first = torch.stack((rows[one], cols[one]), dim=1)
            two = torch.randperm(rows.shape[0])[:m]
            pairs.append([first, torch.stack((rows[two], cols[two]), dim=1)])
    return pairs


class MLP(torch.nn.Module):
    """gaussian_splatting.py:198-213 (`self.fea_up = MLP(self.feature_dim, self.clip_dim, hidden_list=[128])`, :258)"""

    def __init__(self, in_dim=8, out_dim=512, hidden_list=(128,)):
        super().__init__()
        layers, lastv = [], in_dim
        for hidden in hidden_list:
            layers += [torch.nn.Linear(lastv, hidden), torch.nn.ReLU()]
            lastv = hidden
        layers.append(torch.nn.Linear(lastv, out_dim))
        self.layers = torch.nn.Sequential(*layers)

    def forward(self, x):
        return self.layers(x)


class StubCameras:
    """One camera, tensors shaped like nerfstudio's `Cameras[i:i+1]` (`.to(device)` as the datamanager does,
    full_images_datamanager.py:361-372)."""

    def __init__(self, c2w: torch.Tensor, fx: float, fy: float, cx: float, cy: float, height: int, width: int,
                 device="cpu"):
        f = lambda v: torch.tensor([[v]], dtype=torch.float32, device=device)
        self.camera_to_worlds = c2w[None, :3, :].to(device=device, dtype=torch.float32)
        self.fx, self.fy, self.cx, self.cy = f(fx), f(fy), f(cx), f(cy)
        self.width = torch.tensor([[width]], dtype=torch.int64, device=device)
        self.height = torch.tensor([[height]], dtype=torch.int64, device=device)
        self.shape = (1,)
        self.metadata: Optional[Dict] = None      # the datamanager sets {"cam_idx": dataset index} (:375-377)
        self.rescales: List[float] = []

    @property
    def device(self):
        return self.camera_to_worlds.device

    def rescale_output_resolution(self, scaling_factor) -> None:
        """cameras.py:935-961"""
        self.rescales.append(scaling_factor)
        s = torch.tensor([scaling_factor]).to(self.device).broadcast_to(self.cx.shape)
        self.fx, self.fy, self.cx, self.cy = self.fx * s, self.fy * s, self.cx * s, self.cy * s
        self.height = (self.height * s).to(torch.int64)
        self.width = (self.width * s).to(torch.int64)

    @classmethod
    def from_view(cls, view, device="cpu", cam_idx: Optional[int] = None) -> "StubCameras":
        """the nerfstudio / OpenGL camera-to-world of a `camera.ViewParams` (undo the pi rotation about x that
        get_outputs applies, gaussian_splatting.py:658-668)"""
        w2c = view.viewmat.detach().cpu()
        c2w = torch.eye(4)
        c2w[:3, :3] = w2c[:3, :3].T @ torch.diag(torch.tensor([1.0, -1.0, -1.0]))
        c2w[:3, 3] = view.cam_pos.detach().cpu()
        cam = cls(c2w, view.fx, view.fy, view.cx, view.cy, view.height, view.width, device=device)
        if cam_idx is not None:
            cam.metadata = {"cam_idx": int(cam_idx)}
        return cam


class StubCameraOptimizer:
    """CameraOptimizerConfig(mode="off") (gaussian_splatting.py:191): nothing to apply, no parameters"""

    def apply_to_camera(self, camera) -> None:
        return None

    def get_param_groups(self, param_groups: Dict) -> None:
        return None


def default_config(**over) -> types.SimpleNamespace:
    """GaussianSplattingModelConfig defaults (gaussian_splatting.py:150-196)"""
    cfg = dict(warmup_length=500, refine_every=100, resolution_schedule=250, num_downscales=2,
               cull_alpha_thresh=0.1, cull_scale_thresh=0.5, reset_alpha_every=30, densify_grad_thresh=0.0002,
               densify_size_thresh=0.01, n_split_samples=2, sh_degree_interval=1000, cull_screen_size=0.15,
               split_screen_size=0.05, stop_screen_size_at=4000, random_init=False, ssim_lambda=0.2,
               stop_split_at=15000, sh_degree=4, max_gauss_ratio=10.0)
    cfg.update(over)
    return types.SimpleNamespace(**cfg)


class StubGaussianSplattingModel(torch.nn.Module):
    """What `FusedGaussianSplattingModel` and the reference's callbacks use of GaussianSplattingModel."""

    def __init__(self, scene, config: Optional[types.SimpleNamespace] = None, num_train_data: int = 100,
                 step: int = 30000):
        super().__init__()
        for k in ("means", "scales", "quats", "opacities", "colors_all", "feature"):
            setattr(self, k, Parameter(getattr(scene, k).detach().clone()))
        self.config = config or default_config()
        self.num_train_data = num_train_data
        self.step = step
        self.crop_box = None
        self.back_color = torch.zeros(3)
        self.feature_dim = scene.feature.shape[1]
        self.clip_dim = 512
        self.camera_optimizer = StubCameraOptimizer()
        self.xys_grad_norm = None
        self.vis_counts = None
        self.max_2Dsize = None
        self.populate_modules()          # models/base_model.py: Model.__init__ ends with this call

    def populate_modules(self):
        """the part of gaussian_splatting.py:248-299 the plugin's override builds on: `fea_up` (:258)"""
        self.fea_up = MLP(self.feature_dim, self.clip_dim, hidden_list=[128])

    @property
    def device(self):
        return self.means.device

    @property
    def num_points(self) -> int:
        return self.means.shape[0]

    def _get_downscale_factor(self):
        if self.training:
            return 2 ** max((self.config.num_downscales - self.step // self.config.resolution_schedule), 0)
        return 1

    def forward(self, camera):          # models/base_model.py:132-143
        return self.get_outputs(camera)

    def step_cb(self, step):
        self.step = step

    def get_gaussian_param_groups(self) -> Dict[str, List[Parameter]]:
        return {"xyz": [self.means], "color": [self.colors_all], "opacity": [self.opacities],
                "scaling": [self.scales], "rotation": [self.quats], "feature": [self.feature]}

    def get_param_groups(self) -> Dict[str, List[Parameter]]:
        gps = self.get_gaussian_param_groups()
        self.camera_optimizer.get_param_groups(gps)
        return gps


class OrientedBoxStub:
    """`OrientedBox.within` (data/scene_box.py:91-104) for an axis-aligned box: (N, 1) bool like the reference's"""

    def __init__(self, lo, hi):
        self.lo, self.hi = torch.as_tensor(lo, dtype=torch.float32), torch.as_tensor(hi, dtype=torch.float32)

    def within(self, pts: torch.Tensor) -> torch.Tensor:
        lo, hi = self.lo.to(pts.device), self.hi.to(pts.device)
        return ((pts >= lo) & (pts <= hi)).all(dim=-1, keepdim=True)
