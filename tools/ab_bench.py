#!/usr/bin/env python3
"""A/B measurement: run bench.py against ANOTHER build of the library (a kernel variant under test).
Usage: python tools/ab_bench.py path/to/libvariant.so [bench.py flags].  Measurement tool only."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "shim")]
from gaussiangrasper_amd import _lib  # noqa: E402

_lib.LIB_PATH = os.path.abspath(sys.argv[1])
import bench  # noqa: E402

sys.exit(bench.main(sys.argv[2:]))
