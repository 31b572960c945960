"""Synthetic scenes of SURVEY.md §8d (there is no network for scene_0001): seeded, fp32,
the same generator for every BASELINE config."""
from __future__ import annotations

import math
from dataclasses import dataclass

import torch

from .constants import SH_C0

BASE_SEED = 20240309
MEAN_SCALE = 0.0036      # s-bar, world units: sigma_px ~ 2 at z=2.5, fx=1385.6


@dataclass
class Scene:
    """Raw (pre-activation) parameters, named as the reference model's nn.Parameters
    (nerfstudio/models/gaussian_splatting.py:271-281)."""
    means: torch.Tensor        # (N,3)
    scales: torch.Tensor       # (N,3) log-scales
    quats: torch.Tensor        # (N,4) wxyz, normalised
    opacities: torch.Tensor    # (N,1) logits
    colors_all: torch.Tensor   # (N,K,3) SH coefficients
    feature: torch.Tensor      # (N,D)

    def to(self, device):
        return Scene(*[t.to(device) for t in (self.means, self.scales, self.quats, self.opacities,
                                              self.colors_all, self.feature)])

    def params(self):
        return [self.means, self.scales, self.quats, self.opacities, self.colors_all, self.feature]

    @property
    def num_points(self):
        return self.means.shape[0]


def make_scene(num_points: int, feature_dim: int = 32, sh_degree: int = 4, config_index: int = 0,
               max_gauss_ratio: float = 10.0) -> Scene:
    g = torch.Generator(device="cpu").manual_seed(BASE_SEED + config_index)
    n = num_points
    means = (torch.rand(n, 3, generator=g) * 2 - 1) * torch.tensor([1.0, 1.0, 0.5])
    scales = math.log(MEAN_SCALE) + 0.5 * torch.randn(n, 3, generator=g)
    # clamp the axis ratio as the reference's split/regulariser keeps it (max_gauss_ratio, :193)
    smax = scales.max(dim=-1, keepdim=True).values
    scales = torch.maximum(scales, smax - math.log(max_gauss_ratio))
    quats = torch.randn(n, 4, generator=g)
    quats = quats / quats.norm(dim=-1, keepdim=True)
    opacities = 1.5 * torch.randn(n, 1, generator=g)
    k = (sh_degree + 1) ** 2
    colors_all = 0.05 * torch.randn(n, k, 3, generator=g)
    colors_all[:, 0, :] = (torch.rand(n, 3, generator=g) - 0.5) / SH_C0   # RGB2SH (:73-78)
    feature = torch.rand(n, feature_dim, generator=g) * 2 - 1
    return Scene(means.float(), scales.float(), quats.float(), opacities.float(),
                 colors_all.float(), feature.float())
