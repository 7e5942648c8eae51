#!/usr/bin/env python3
"""A/B timing of one scene against another build of the library: python tools/ab_rates.py LIB [scene] [spp]
prints the frame time and the extend / shade kernel times (RT_FLAG_TIMING) of a second frame."""
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
import raytracing_amd as rt

name = sys.argv[2] if len(sys.argv) > 2 else "cu_cover2"
spp = int(sys.argv[3]) if len(sys.argv) > 3 else 1000
world, kw = rt.get_scene(name, spp=spp)
dev = rt.DeviceScene(world)
args = (kw["camera"], kw["pattern"], kw["background"], kw["width"], kw["height"])
dev.render(*args)
best = None
for _ in range(3):
    img, st = dev.render(*args, flags=2)
    if best is None or st.seconds < best.seconds:
        best = st
paths = kw["width"] * kw["height"] * spp
print("%-40s %s %d spp: %.4f s %.1f Mpaths/s | extend %.4f shade %.4f" % (os.path.basename(sys.argv[1]), name, spp, best.seconds,
      paths / best.seconds / 1e6, best.extend_seconds, best.shade_seconds), flush=True)
