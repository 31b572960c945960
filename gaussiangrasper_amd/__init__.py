"""gaussiangrasper_amd — MI355X-native differentiable Gaussian-splatting rasterizer for
GaussianGrasper's feature-field hot path (project -> tile bin/sort -> alpha-blend fwd/bwd).

Public surface = the gsplat-0.1.0 operators the reference model imports
(nerfstudio/models/gaussian_splatting.py:46-50); `shim/gsplat` re-exports them under the
module paths the reference uses, so the reference files run unchanged with
PYTHONPATH=<repo>/shim:<repo>."""
from .constants import num_sh_bases  # noqa: F401
from .ops import (NDRasterizeGaussians, ProjectGaussians, RasterizeGaussians,  # noqa: F401
                  SphericalHarmonics, bin_and_sort_gaussians, quat_to_rotmat)

__all__ = ["ProjectGaussians", "SphericalHarmonics", "RasterizeGaussians", "NDRasterizeGaussians",
           "quat_to_rotmat", "num_sh_bases", "bin_and_sort_gaussians"]
