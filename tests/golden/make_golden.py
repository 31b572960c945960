"""Generates the golden fixtures in this directory FROM THE CPU ORACLE (oracle/gg_oracle.c, f32
build).  The reference holds no fixture for this path and its implementation (gsplat==0.1.0) is not
available offline (parity unpinned), so these vectors pin the oracle against
regressions and give the GPU box (which has no /root/reference and needs none) known answers.

    python tests/golden/make_golden.py          # rewrites tests/golden/*.npz
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from gaussiangrasper_amd.camera import ring_cameras  # noqa: E402
from gaussiangrasper_amd.scene import make_scene  # noqa: E402
from oracle import oracle as O  # noqa: E402

CASES = [  # name, N, H, W, C, scale multiplier (bigger splats at tiny resolutions), view
    ("g0_n7_45x70_c3", 7, 45, 70, 3, 6.0, 0),
    ("g1_n300_48x64_c3", 300, 48, 64, 3, 14.0, 1),
    ("g2_n200_32x48_c8", 200, 32, 48, 8, 5.0, 2),
    ("g3_n150_16x32_c32", 150, 16, 32, 32, 12.0, 0),
    ("g4_dense_n400_32x32_c3", 400, 32, 32, 3, 60.0, 1),   # clustered + opaque: hits the T<=1e-4 stop
]


def main():
    O.build()
    for name, n, h, w, ch, smul, vi in CASES:
        sc = make_scene(n, feature_dim=ch, config_index=100 + len(name))
        v = ring_cameras(3, h, w)[vi]
        scales = (sc.scales.exp() * smul).numpy()
        if "dense" in name:
            sc.means.mul_(0.5)
            sc.opacities.add_(6.0)
        viewmat = v.viewmat[:3].contiguous().numpy()
        projmat = v.projmat.numpy()
        means, quats = sc.means.numpy(), sc.quats.numpy()
        xys, depths, radii, conics, nth, cov3d = O.project_fwd(
            means, scales, 1.0, quats, viewmat, projmat, v.fx, v.fy, v.cx, v.cy, h, w, v.tile_bounds)
        rng = np.random.default_rng(len(name))
        colors = rng.uniform(-1, 1, (n, ch)).astype(np.float32)
        opacity = torch.sigmoid(sc.opacities).numpy()
        background = rng.uniform(0, 1, ch).astype(np.float32)
        out, saved = O.rasterize_fwd(xys, depths, radii, conics, nth, colors, opacity, h, w, background)
        b = saved["bins"]
        assert b["num_intersects"] > 0
        v_out = rng.standard_normal(out.shape).astype(np.float32)
        v_xy, v_conic, v_colors, v_opacity = O.blend_bwd(
            b["gaussian_ids_sorted"], b["tile_bins"], xys, conics, colors, opacity, h, w, background,
            saved["final_Ts"], saved["final_idx"], v_out)
        v_mean3d, v_scale, v_quat = O.project_bwd(
            means, scales, 1.0, quats, viewmat, projmat, v.fx, v.fy, v.cx, v.cy, h, w, radii, conics,
            v_xy, np.zeros(n, np.float32), v_conic)
        np.savez_compressed(
            os.path.join(HERE, name + ".npz"), means=means, scales=scales, quats=quats,
            viewmat=viewmat, projmat=projmat, intr=np.array([v.fx, v.fy, v.cx, v.cy], np.float64),
            hw=np.array([h, w]), xys=xys, depths=depths, radii=radii, conics=conics,
            num_tiles_hit=nth, cov3d=cov3d, isect_ids_sorted=b["isect_ids_sorted"],
            gaussian_ids_sorted=b["gaussian_ids_sorted"], tile_bins=b["tile_bins"], colors=colors,
            opacity=opacity, background=background, out_img=out, final_Ts=saved["final_Ts"],
            final_idx=saved["final_idx"], v_out=v_out, v_xy=v_xy, v_conic=v_conic,
            v_colors=v_colors, v_opacity=v_opacity, v_mean3d=v_mean3d, v_scale=v_scale,
            v_quat=v_quat)
        if "dense" in name:
            assert (saved["final_Ts"] < 1e-3).sum() > 50, "dense case must exercise the early stop"
        print(name, "I =", b["num_intersects"], "visible =", int((radii > 0).sum()),
              "mean final_T = %.3f" % saved["final_Ts"].mean())


if __name__ == "__main__":
    main()
