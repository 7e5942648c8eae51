#!/usr/bin/env python3
"""A/B measurement: bench.py run against another build of the library (python tools/ab_bench.py LIB [bench.py flags]).
Symbols the other build lacks are left out of the binding table."""
import ctypes
import os
import sys

import torch  # noqa: F401

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from raytracing_amd import _ffi

_ffi.LIB_PATH = os.path.abspath(sys.argv[1])
probe = ctypes.CDLL(_ffi.LIB_PATH)
for name in list(_ffi.SYMBOLS):
    if not hasattr(probe, name):
        del _ffi.SYMBOLS[name]
sys.argv = ["bench.py"] + sys.argv[2:]
import bench

bench.main()
