#!/usr/bin/env python3
"""Mpaths/s of whole frames (second frame, no timing events) per scene, pattern columns and slot-pool size.
python tools/pool_sizes.py scene:spp[,scene:spp...] slots[,slots...]   (slots in Mi; 0 = the library's own choice)"""
import os
import sys
import time

import torch  # noqa: F401

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import raytracing_amd as rt  # noqa: E402

cases = [c.split(":") for c in sys.argv[1].split(",")]
pools = [int(x) for x in sys.argv[2].split(",")]
for name, spp in cases:
    spp = int(spp)
    world, kw = rt.get_scene(name, spp=spp)
    args = (kw["camera"], kw["pattern"], kw["background"], kw["width"], kw["height"])
    paths = kw["width"] * kw["height"] * spp
    row = []
    for mi in pools:
        if mi:
            os.environ["RT_AMD_SLOTS"] = str(mi << 20)
        else:
            os.environ.pop("RT_AMD_SLOTS", None)
        dev = rt.DeviceScene(world)
        dev.render(*args)
        best = 1e9
        for _ in range(2):
            t0 = time.perf_counter()
            dev.render(*args)
            best = min(best, time.perf_counter() - t0)
        dev.close()
        row.append("%s %7.1f" % ("auto" if mi == 0 else "%2d Mi" % mi, paths / best / 1e6))
    print("%-18s %5d spp %6.0f M paths | %s" % (name, spp, paths / 1e6, " | ".join(row)), flush=True)
