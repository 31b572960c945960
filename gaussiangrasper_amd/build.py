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
SOURCES = ["project.hip", "binning.hip", "blend.hip", "blend2.hip", "mlp.hip", "prof.hip"]
FLAGS = ["-O3", "-fPIC", "-shared", "--offload-arch=gfx950", "-ffp-contract=off", "-std=c++17",
         "-Wall", "-Wno-unused-function"]


def _stale() -> bool:
    if not os.path.exists(OUT):
        return True
    t = os.path.getmtime(OUT)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)]
    deps += [os.path.join(HERE, "..", "include", f) for f in ("gg_raster.h", "gg_constants.h")]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = False, extra_flags=()) -> str:
    if not force and not _stale():
        return OUT
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    cmd = [hipcc, *FLAGS, *extra_flags, *[os.path.join(CSRC, s) for s in SOURCES], "-o", OUT]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return OUT


if __name__ == "__main__":
    build(force="--force" in sys.argv, verbose="--verbose" in sys.argv or True)
    print(OUT)
