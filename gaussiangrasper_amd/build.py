"""Builds libgg_raster.so (the C-ABI HIP library, include/gg_raster.h) for gfx950, in-tree.

    python -m gaussiangrasper_amd.build [--force] [--verbose]

hipcc cross-compiles without a GPU.  -ffp-contract=off is part of the numerical contract:
the forward kernels must round exactly like the oracle (no implicit FMA fusion)."""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT = os.path.join(HERE, "libgg_raster.so")
SOURCES = ["project.hip", "binning.hip", "blend.hip", "blend2.hip", "mlp.hip", "losses.hip", "imgloss.hip", "densify.hip", "prof.hip"]
# -fno-slp-vectorize: hipcc's SLP pass packs adjacent scalar fp32 operations into v_pk_*_f32 (15 % of the
# blend loops' VALU instructions); on gfx950 a packed fp32 instruction is not cheaper than its two halves
# (MI355X_MICROARCH.md "packed f32 VALU ... an anti-lever") and the packing costs 8-12 VGPRs per kernel, i.e.
# one or two resident waves per SIMD.  Same operations, same rounding: results are unchanged.
FLAGS = ["-O3", "-fPIC", "-shared", "--offload-arch=gfx950", "-ffp-contract=off", "-fno-slp-vectorize",
         "-std=c++17", "-Wall", "-Wno-unused-function"]


def _stale(out: str = None) -> bool:
    out = out or OUT
    if not os.path.exists(out):
        return True
    t = os.path.getmtime(out)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)]
    deps += [os.path.join(HERE, "..", "include", f) for f in ("gg_raster.h", "gg_constants.h")]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = False, extra_flags=(), out: str = OUT) -> str:
    if not force and out == OUT and not _stale():
        return out
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    cmd = [hipcc, *FLAGS, *extra_flags, *[os.path.join(CSRC, s) for s in SOURCES], "-o", out]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return out


ABL_OUT = os.path.join(HERE, "libgg_raster_abl.so")


def build_ablation(verbose: bool = False, walk_stats: bool = False) -> str:
    """Measurement-only twin of the library (tools/kbench.py): the same sources with -DGG_ABLATION,
    which adds the ablated kernels and the `gg_debug_*` switches; walk_stats (tools/walkstats.py) also counts
    what the forward walks (the counters make that kernel ~100x slower).  Never loaded by the product."""
    flags = ("-DGG_ABLATION", "-DGG_WALK_STATS") if walk_stats else ("-DGG_ABLATION",)
    return build(force=True, verbose=verbose, extra_flags=flags, out=ABL_OUT)


COMPAT_OUT = os.path.join(HERE, "libgg_raster_compat.so")
COMPAT_FLAGS = ("-DGG_VJP_GSPLAT_COMPAT=1", "-DGG_ALPHA_MAX_BWD=0.99f")


def build_compat(verbose: bool = False, force: bool = True) -> str:
    """Variant with the recalled gsplat-0.1.0 deviations switched on (include/gg_constants.h, PARITY.md):
    tests/test_compat_variant.py holds it to the oracle built with the same switches.  Not the product.
    force=False rebuilds only when a source is newer than the library."""
    if not force and not _stale(COMPAT_OUT):
        return COMPAT_OUT
    return build(force=True, verbose=verbose, extra_flags=COMPAT_FLAGS, out=COMPAT_OUT)


if __name__ == "__main__":
    if "--compat" in sys.argv:
        print(build_compat(verbose=True))
    elif "--ablation" in sys.argv:
        print(build_ablation(verbose=True))
    else:
        build(force="--force" in sys.argv, verbose="--verbose" in sys.argv or True)
        print(OUT)
        if "--all" in sys.argv:
            print(build_compat(verbose=True))
