"""Python mirror of include/gg_constants.h (the †UNVERIFIED gsplat-0.1.0
constants, kept in one header + this one module; tests/test_constants.py checks they agree)."""

CLIP_THRESH_DEFAULT = 0.01
BLUR = 0.3
FOV_LIM = 1.3
RADIUS_SIGMA = 3.0
EIG_FLOOR = 0.1
W_EPS = 1e-6
PIX_OFFSET = 0.5
BLOCK = 16
ALPHA_MAX_FWD = 0.999
ALPHA_MAX_BWD = 0.999
ALPHA_MIN = 1.0 / 255.0
T_EPS = 1e-4
SH_C0 = 0.28209479177387814
SH_MAX_BASES = 25


def num_sh_bases(degree: int) -> int:
    """gsplat.sh.num_sh_bases (reference import nerfstudio/models/gaussian_splatting.py:50):
    0->1, 1->4, 2->9, 3->16, >=4->25."""
    if degree == 0:
        return 1
    if degree == 1:
        return 4
    if degree == 2:
        return 9
    if degree == 3:
        return 16
    return 25


def deg_from_sh(num_bases: int) -> int:
    """Inverse of num_sh_bases for the stored coefficient count."""
    for d, n in ((0, 1), (1, 4), (2, 9), (3, 16), (4, 25)):
        if num_bases == n:
            return d
    raise ValueError(f"invalid number of SH bases: {num_bases}")
