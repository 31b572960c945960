"""No-GPU checks of the drop-in boundary: the C-ABI library loads and exports every symbol that
include/gg_raster.h declares (no compute calls here), the ctypes table mirrors the header, and the
†constants agree between the header, the Python module and the oracle build."""
import ctypes
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_functions():
    src = open(os.path.join(ROOT, "include", "gg_raster.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(gg_[a-z0-9_]+)\s*\(", src)))


def test_header_declares_the_expected_surface():
    names = _header_functions()
    for required in ("gg_project_fwd", "gg_project_bwd", "gg_sh_fwd", "gg_sh_bwd", "gg_bin_sort",
                     "gg_blend_fwd", "gg_blend_bwd", "gg_last_error", "gg_count_intersects"):
        assert required in names


def test_library_exports_every_declared_symbol():
    from gaussiangrasper_amd import _lib
    lib = _lib.load()  # builds with hipcc if the .so is absent (cross-compiles without a GPU)
    raw = ctypes.CDLL(_lib.LIB_PATH)
    for name in _header_functions():
        assert hasattr(raw, name), f"{name} declared in gg_raster.h but not exported"
        assert name in _lib.SIGNATURES, f"{name} missing from the ctypes table"
    assert sorted(_lib.SIGNATURES) == _header_functions()
    assert lib.gg_abi_version() == _lib.ABI_VERSION
    assert lib.gg_last_error() == b""


def test_workspace_queries_are_pure_host_calls():
    from gaussiangrasper_amd import _lib
    lib = _lib.load()
    assert lib.gg_blend_workspace(1000) >= 32 * 1000
    assert lib.gg_blend_workspace(0) > 0
    small, big = lib.gg_bin_sort_workspace(1000, 5000), lib.gg_bin_sort_workspace(1_000_000, 4_000_000)
    assert 0 < small < big < 200 * 2 ** 20   # 1M / 4M intersections needs < 200 MiB of scratch
    # the fast MLP forward: widest out_dim its LDS holds is 3968 (two 64 KB weight slices + biases in 160 KB)
    from gaussiangrasper_amd.mlp import FAST_MAX_OUT
    assert FAST_MAX_OUT == 3968
    assert lib.gg_mlp_fwd_fast_workspace(128, 128, 512) > 0 and lib.gg_mlp_fwd_fast_workspace(128, 128, FAST_MAX_OUT) > 0
    assert lib.gg_mlp_fwd_fast_workspace(128, 128, FAST_MAX_OUT + 16) == 0
    n = ctypes.c_void_p(0)
    assert lib.gg_mlp_fwd_fast(1, 128, 128, 4096, n, n, n, n, n, n, n, 0, n) == -1 and b"3968" in lib.gg_last_error()


def test_argument_validation_without_a_gpu():
    """invalid arguments are rejected on the host before anything is launched"""
    from gaussiangrasper_amd import _lib
    lib = _lib.load()
    n = ctypes.c_void_p(0)
    st = lib.gg_project_fwd(-1, n, n, 1.0, n, n, n, 1.0, 1.0, 0.0, 0.0, 16, 16, 1, 1, 0.01, n, n, n,
                            n, n, n, n)
    assert st == -1 and b"num_points" in lib.gg_last_error()
    st = lib.gg_project_fwd(4, n, n, 1.0, n, n, n, 1.0, 1.0, 0.0, 0.0, 16, 16, 7, 1, 0.01, n, n, n,
                            n, n, n, n)
    assert st == -1 and b"tile_bounds" in lib.gg_last_error()
    st = lib.gg_sh_fwd(4, 5, 1, n, n, n, n)
    assert st == -1 and b"num_bases" in lib.gg_last_error()
    # the depth sort carries a Gaussian's tile box in one 32-bit word: grids beyond 1023 x 1023 are refused, not mis-sorted
    st = lib.gg_bin_sort(4, 4, n, n, n, n, 1024, 2, n, n, n, n, 0, n)
    assert st == -1 and b"1023" in lib.gg_last_error()
    st = lib.gg_sh_fwd(4, 4, 2, n, n, n, n)
    assert st == -1 and b"degrees_to_use" in lib.gg_last_error()
    st = lib.gg_blend_fwd(0, 4, 16, 16, n, n, n, n, n, n, n, n, n, n, n, 0, n)
    assert st == -1
    with pytest.raises(_lib.GGError):
        _lib.check(st, "gg_blend_fwd")


def test_constants_agree_between_header_and_python():
    from gaussiangrasper_amd import constants as K
    src = open(os.path.join(ROOT, "include", "gg_constants.h")).read()
    defs = dict(re.findall(r"#define\s+(GG_[A-Z0-9_]+)\s+([-0-9.ef()/ ]+?)\s*(?:/\*|$)", src, flags=re.M))
    val = lambda s: float(eval(s.replace("f", "")))  # noqa: S307 - header literals only
    assert val(defs["GG_CLIP_THRESH_DEFAULT"]) == K.CLIP_THRESH_DEFAULT
    assert val(defs["GG_BLUR"]) == K.BLUR and val(defs["GG_FOV_LIM"]) == K.FOV_LIM
    assert val(defs["GG_RADIUS_SIGMA"]) == K.RADIUS_SIGMA and val(defs["GG_EIG_FLOOR"]) == K.EIG_FLOOR
    assert val(defs["GG_W_EPS"]) == K.W_EPS and val(defs["GG_PIX_OFFSET"]) == K.PIX_OFFSET
    assert int(val(defs["GG_BLOCK"])) == K.BLOCK
    assert val(defs["GG_ALPHA_MAX_FWD"]) == K.ALPHA_MAX_FWD and val(defs["GG_ALPHA_MAX_BWD"]) == K.ALPHA_MAX_BWD
    assert abs(val(defs["GG_ALPHA_MIN"]) - K.ALPHA_MIN) < 1e-12 and val(defs["GG_T_EPS"]) == K.T_EPS
    assert val(defs["GG_SH_C0"]) == K.SH_C0


def test_num_sh_bases_map():
    from gsplat.sh import num_sh_bases
    assert [num_sh_bases(d) for d in range(6)] == [1, 4, 9, 16, 25, 25]


def test_sh_constants_match_in_tree_magnitudes():
    """the reference tree's own SH basis (nerfstudio/utils/math.py:54-90, all-positive signs) has
    the same MAGNITUDES as the gsplat table restated in gg_constants.h"""
    src = open(os.path.join(ROOT, "include", "gg_constants.h")).read()
    mags = sorted({round(abs(float(v.rstrip("f"))), 12) for v in
                   re.findall(r"#define\s+GG_SH_C[0-4](?:_\d)?\s+(-?[0-9.]+f)", src)})
    in_tree = sorted({round(x, 12) for x in (
        0.28209479177387814, 0.4886025119029199, 1.0925484305920792, 0.31539156525252005 * 1,
        0.5462742152960396, 0.5900435899266435, 2.890611442640554, 0.4570457994644658,
        0.3731763325901154, 1.445305721320277, 2.5033429417967046, 1.7701307697799304,
        0.9461746957575601, 0.6690465435572892, 0.10578554691520431, 0.47308734787878004,
        0.6258357354491761)})
    assert mags == in_tree
