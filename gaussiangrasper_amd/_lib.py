"""ctypes binding of libgg_raster.so (C ABI: include/gg_raster.h).

There is NO CPU fallback: if the library is missing and cannot be built, or an operator is
handed a non-HIP tensor, the call raises.  (The CPU restatement under oracle/ is test
infrastructure and is never imported from here.)"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libgg_raster.so")
ABI_VERSION = 4

_P, _I, _F, _I64, _SZ = C.c_void_p, C.c_int, C.c_float, C.c_int64, C.c_size_t


class RowArray(C.Structure):
    """gg_row_array_t"""
    _fields_ = [("src", C.c_void_p), ("dst", C.c_void_p), ("row_floats", C.c_int), ("kind", C.c_int)]


class AdamGroup(C.Structure):
    """gg_adam_group_t"""
    _fields_ = [("param", C.c_void_p), ("grad", C.c_void_p), ("exp_avg", C.c_void_p),
                ("exp_avg_sq", C.c_void_p), ("numel", C.c_int64), ("lr", C.c_double),
                ("beta1", C.c_double), ("beta2", C.c_double), ("eps", C.c_double),
                ("weight_decay", C.c_double), ("step", C.c_int64)]


ROWS_COPY, ROWS_MEANS, ROWS_SCALES, ROWS_ZERO_NEW = 0, 1, 2, 3
MAX_ROW_ARRAYS, ADAM_MAX_GROUPS = 24, 8

# name -> (restype, argtypes); mirrors include/gg_raster.h declaration by declaration
SIGNATURES = {
    "gg_abi_version": (_I, []),
    "gg_last_error": (C.c_char_p, []),
    "gg_project_fwd": (_I, [_I, _P, _P, _F, _P, _P, _P, _F, _F, _F, _F, _I, _I, _I, _I, _F,
                            _P, _P, _P, _P, _P, _P, _P]),
    "gg_project_count_workspace": (_SZ, [_I]),
    "gg_project_fwd_count": (_I, [_I, _P, _P, _F, _P, _P, _P, _F, _F, _F, _F, _I, _I, _I, _I, _F,
                                  _P, _P, _P, _P, _P, _P, _P, _P, _SZ, _P]),
    "gg_project_bwd": (_I, [_I, _P, _P, _F, _P, _P, _P, _F, _F, _F, _F, _I, _I, _P, _P, _P, _P, _P,
                            _P, _P, _P, _P]),
    "gg_project_bwd_ex": (_I, [_I, _P, _P, _F, _P, _P, _P, _F, _F, _F, _F, _I, _I, _P, _P, _P, _I, _P, _P, _I,
                               _P, _I, _P, _P, _P]),
    "gg_sh_fwd": (_I, [_I, _I, _I, _P, _P, _P, _P]),
    "gg_sh_bwd": (_I, [_I, _I, _I, _P, _P, _P, _P]),
    "gg_sh_bwd_accumulate": (_I, [_I, _I, _I, _P, _P, _P, _P]),
    "gg_quat_to_rotmat_fwd": (_I, [_I, _P, _P, _P]),
    "gg_quat_to_rotmat_bwd": (_I, [_I, _P, _P, _P, _P]),
    "gg_activate_fwd": (_I, [_I] + [_P] * 12),
    "gg_activate_bwd": (_I, [_I] + [_P] * 12),
    "gg_activate_bwd_ex": (_I, [_I] + [_P] * 7 + [_I] + [_P] * 4 + [_I, _P]),
    "gg_mlp_fwd": (_I, [_I64, _I, _I, _I, _P, _P, _P, _P, _P, _P, _P]),
    "gg_debug_set_fwd_blocks": (_I, [_I, _I]),
    "gg_mlp_fwd_fast_workspace": (_SZ, [_I, _I, _I]),
    "gg_mlp_fwd_fast": (_I, [_I64, _I, _I, _I, _P, _P, _P, _P, _P, _P, _P, _SZ, _P]),
    "gg_mlp_bwd": (_I, [_I64, _I, _I, _I] + [_P] * 11),
    "gg_cosine_loss_fwd": (_I, [_I64, _I, _P, _P, _P, _P, _P, _P, _P]),
    "gg_cosine_loss_bwd": (_I, [_I64, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    "gg_count_workspace": (_SZ, [_I]),
    "gg_count_intersects": (_I, [_I, _P, _P, _P, _SZ, _P]),
    "gg_bin_sort_workspace": (_SZ, [_I, _I64]),
    "gg_bin_sort": (_I, [_I, _I64, _P, _P, _P, _P, _I, _I, _P, _P, _P, _P, _SZ, _P]),
    "gg_bin_sort_status": (_I, [_I, _I64, _P, _SZ, _P]),
    "gg_bin_sort_dev": (_I, [_I, _I64, _P, _P, _P, _P, _P, _I, _I, _P, _P, _P, _P, _SZ, _P]),
    "gg_bin_sort_dev_ex": (_I, [_I, _I64, _P, _P, _P, _P, _P, _I, _I, _P, _P, _P, _P, _SZ, _P, _P, _I, _P]),
    "gg_view_fwd_workspace": (_SZ, [_I]),
    "gg_view_fwd": (_I, [_I, _P, _P, _P, _P, _P, _P, _P, _F, _F, _F, _F, _I, _I, _I, _I, _F] + [_P] * 11 +
                    [_P, _P, _SZ, _P, _SZ, _P]),
    "gg_blend_fwd_pair_packed": (_I, [_I, _I, _I, _I, _I] + [_P] * 11 + [_SZ, _I, _P]),
    "gg_blend_workspace": (_SZ, [_I]),
    "gg_blend_fwd": (_I, [_I, _I, _I, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _SZ, _P]),
    "gg_blend_fwd_pair": (_I, [_I, _I, _I, _I, _I] + [_P] * 14 + [_SZ, _P]),
    "gg_blend_fwd_pair_fast": (_I, [_I, _I, _I, _I, _I] + [_P] * 14 + [_SZ, _P]),
    "gg_blend_bwd": (_I, [_I, _I, _I, _I] + [_P] * 14 + [_I, _I, _P, _SZ, _I, _P]),
    "gg_shade_tail_bwd_split": (_I, [_I, _P, _I, _P, _P, _P, _P, _P]),
    "gg_view_bwd": (_I, [_I, _P, _I, _P, _P, _P, _F, _P, _P, _P, _P, _P, _P, _F, _F, _I, _I, _P, _P, _P, _P, _P, _P, _P, _P]),
    "gg_sh_bwd_multi": (_I, [_I, _I, _I, _I, C.POINTER(_P), C.POINTER(_P), _P, _I, _P]),
    "gg_image_loss_workspace": (_SZ, [_I, _I]),
    "gg_image_loss_fwd": (_I, [_I, _I, _P, _I, _P, _P, _F, _P, _P, _SZ, _P]),
    "gg_image_loss_bwd": (_I, [_I, _I, _P, _I, _P, _P, _F, _P, _P, _SZ, _P, _P]),
    "gg_geom_loss_workspace": (_SZ, []),
    "gg_geom_loss_fwd": (_I, [_I64, _P, _I, _P, _I, _P, _I, _I, _P, _I, _I, _P, _P, _P, _SZ, _P]),
    "gg_geom_loss_bwd": (_I, [_I64, _P, _I, _P, _I, _P, _I, _I, _P, _I, _I, _P, _P, _P, _P, _SZ, _P, _P, _P]),
    "gg_shade_tail_fwd": (_I, [_I, _I, _I, _P, _P, _P, _P, _P, _P, _P]),
    "gg_shade_tail_bwd": (_I, [_I, _I, _I, _P, _P, _I, _P, _P, _I, _P, _P, _P]),
    "gg_blend_bwd_pair": (_I, [_I, _I, _I, _I, _I] + [_P] * 12 + [C.POINTER(_P), C.POINTER(_I), _I] + [_P] * 5
                          + [_I, _I, _I, _P, _SZ, _I, _P]),
    "gg_blend_bwd_deterministic_workspace": (_SZ, [_I, _I, _I64]),
    "gg_blend_bwd_deterministic": (_I, [_I, _I, _I, _I] + [_P] * 14 + [_I, _I, _P, _SZ, _I, _I64, _P, _SZ, _P]),
    "gg_expf_array": (_I, [_I, _P, _P, _P]),
    "gg_rows_workspace": (_SZ, [_I]),
    "gg_mask_scan": (_I, [_I, _P, _I, _P, _P, _P, _SZ, _P]),
    "gg_compact_rows": (_I, [_I, _P, _I, C.POINTER(RowArray), _P, _P, _SZ, _P]),
    "gg_densify_rows": (_I, [_I, _P, _P, _P, _P, _I, _I, _I, _P, _F, _P, _P, _P, _I,
                             C.POINTER(RowArray), _P]),
    "gg_densify_stats": (_I, [_I, _P, _P, _I, _I, _P, _P, _P, _P]),
    "gg_densify_masks": (_I, [_I, _P, _P, _P, _P, _I, _F, _F, _F, _I, _F, _P, _P, _P]),
    "gg_cull_mask": (_I, [_I, _P, _P, _P, _F, _F, _F, _I, _I, _P, _P]),
    "gg_adam_step": (_I, [_I, C.POINTER(AdamGroup), _I, _P]),
    "gg_prof_enable": (_I, [_I]),
    "gg_prof_reset": (_I, []),
    "gg_prof_get": (_I, [_I, C.POINTER(C.c_int), C.POINTER(C.c_double)]),
    "gg_prof_name": (C.c_char_p, [_I]),
}

_lib = None


class GGError(RuntimeError):
    pass


def load(build_if_missing: bool = True):
    """Load (building first if the .so is absent and hipcc is available) and type the library."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        if not build_if_missing:
            raise GGError(f"{LIB_PATH} not found; run `python -m gaussiangrasper_amd.build`")
        from . import build as _build
        try:
            _build.build()
        except Exception as exc:  # noqa: BLE001 - surface the build failure loudly
            raise GGError(f"libgg_raster.so is missing and could not be built: {exc}") from exc
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the .so does not export the symbol
        fn.restype, fn.argtypes = res, args
    ver = lib.gg_abi_version()
    if ver != ABI_VERSION:
        raise GGError(f"libgg_raster.so ABI {ver} != expected {ABI_VERSION}; rebuild")
    _lib = lib
    return lib


def load_variant(path: str):
    """A separately typed handle on a VARIANT build of the library (tests / tools only: the compat and
    ablation twins).  Never becomes the library the operators use."""
    lib = C.CDLL(path)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype, fn.argtypes = res, args
    return lib


def check(status: int, what: str) -> None:
    if status != 0:
        msg = load().gg_last_error().decode("utf-8", "replace")
        raise GGError(f"{what} failed ({status}): {msg}")
