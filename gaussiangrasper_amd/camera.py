"""Camera preparation the reference model does before it calls the rasterizer
(nerfstudio/models/gaussian_splatting.py:655-682), written for plain
4x4 camera-to-world matrices so the bench / tests can drive the operators exactly as
`get_outputs` does without nerfstudio's `Cameras` class."""
from __future__ import annotations

import math

import numpy as np
from dataclasses import dataclass

import torch

from .constants import BLOCK


def projection_matrix(znear: float, zfar: float, fovx: float, fovy: float, device="cpu") -> torch.Tensor:
    """OpenGL-style perspective matrix with w_clip = z_view; same entries as the reference's
    `projection_matrix` (gaussian_splatting.py:87-105)."""
    top = znear * math.tan(0.5 * fovy)
    right = znear * math.tan(0.5 * fovx)
    m = torch.zeros(4, 4, dtype=torch.float32)
    m[0, 0] = 2.0 * znear / (2.0 * right)
    m[1, 1] = 2.0 * znear / (2.0 * top)
    m[2, 2] = (zfar + znear) / (zfar - znear)
    m[2, 3] = -1.0 * zfar * znear / (zfar - znear)
    m[3, 2] = 1.0
    return m.to(device)


@dataclass
class ViewParams:
    """What one call of the operator sequence needs: viewmat (4,4), full projmat (4,4),
    intrinsics, image size and tile bounds."""
    viewmat: torch.Tensor
    projmat: torch.Tensor
    fx: float
    fy: float
    cx: float
    cy: float
    height: int
    width: int
    cam_pos: torch.Tensor

    @property
    def tile_bounds(self):
        return ((self.width + BLOCK - 1) // BLOCK, (self.height + BLOCK - 1) // BLOCK, 1)


def view_from_c2w(c2w: torch.Tensor, fx: float, fy: float, cx: float, cy: float, height: int,
                  width: int, device="cpu") -> ViewParams:
    """c2w: (3,4) or (4,4) nerfstudio/OpenGL camera-to-world.  Mirrors :661-676: rotate pi about
    x (diag(1,-1,-1)), analytic inverse, fov from intrinsics, projmat(0.001, 1000)."""
    c2w = c2w.to(torch.float32).cpu()
    R = c2w[:3, :3] @ torch.diag(torch.tensor([1.0, -1.0, -1.0]))
    T = c2w[:3, 3:4]
    R_inv = R.T
    T_inv = -R_inv @ T
    viewmat = torch.eye(4, dtype=torch.float32)
    viewmat[:3, :3] = R_inv
    viewmat[:3, 3:4] = T_inv
    # :672-673 `2 * math.atan(camera.width / (2 * camera.fx))`: the quotient is formed from tensors (int64 by
    # float32 -> float32), the arc tangent in double
    f32 = np.float32
    fovx = 2 * math.atan(float(f32(width) / (f32(2.0) * f32(fx))))
    fovy = 2 * math.atan(float(f32(height) / (f32(2.0) * f32(fy))))
    projmat = projection_matrix(0.001, 1000, fovx, fovy)
    return ViewParams(viewmat.to(device), (projmat @ viewmat).to(device), float(fx), float(fy),
                      float(cx), float(cy), int(height), int(width), c2w[:3, 3].clone().to(device))


def ring_cameras(num_views: int, height: int, width: int, radius: float = 2.5,
                 elevation_deg: float = 30.0, fov_x_deg: float = 60.0, device="cpu"):
    """SURVEY.md §8d cameras: V views on a ring looking at the origin, world z up, OpenGL camera
    axes (x right, y up, looking down -z)."""
    fx = fy = 0.5 * width / math.tan(math.radians(fov_x_deg) / 2)
    cx, cy = width / 2.0, height / 2.0
    views = []
    el = math.radians(elevation_deg)
    for v in range(num_views):
        az = 2 * math.pi * v / max(num_views, 1)
        pos = torch.tensor([radius * math.cos(el) * math.cos(az),
                            radius * math.cos(el) * math.sin(az),
                            radius * math.sin(el)], dtype=torch.float64)
        back = pos / pos.norm()                      # camera +z points away from the target
        up = torch.tensor([0.0, 0.0, 1.0], dtype=torch.float64)
        right = torch.linalg.cross(up, back)
        right = right / right.norm()
        cam_up = torch.linalg.cross(back, right)
        c2w = torch.eye(4, dtype=torch.float64)
        c2w[:3, 0], c2w[:3, 1], c2w[:3, 2], c2w[:3, 3] = right, cam_up, back, pos
        views.append(view_from_c2w(c2w.float(), fx, fy, cx, cy, height, width, device))
    return views
