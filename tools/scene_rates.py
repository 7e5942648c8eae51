#!/usr/bin/env python3
"""Every sample scene once: Mpaths/s of a second frame (buffers warm) and the walk statistics of the counting build.
python tools/scene_rates.py [spp scale]"""
import os
import sys

import torch  # noqa: F401

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import raytracing_amd as rt

scale = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
for name, spp in (("cu_cover2", 1000), ("cornell_box", 1000), ("cornell_smoke_box", 1000), ("cover1", 500), ("cover2", 300), ("teapot", 1000)):
    spp = max(1, int(spp * scale))
    world, kw = rt.get_scene(name, spp=spp)
    dev = rt.DeviceScene(world)
    args = (kw["camera"], kw["pattern"], kw["background"], kw["width"], kw["height"])
    dev.render(*args)
    img, st = dev.render(*args, flags=2)
    paths = kw["width"] * kw["height"] * spp
    _, ct = dev.render(kw["camera"], kw["pattern"][:, :max(1, spp // 20)], kw["background"], kw["width"], kw["height"], flags=1)
    print("%-18s %5d spp %8.4f s %8.1f Mpaths/s | extend %.4f shade %.4f s | rays/path %.2f nodes/ray %.1f prims/ray %.2f lanes: step %.2f leaf %.2f" % (
        name, spp, st.seconds, paths / st.seconds / 1e6, st.extend_seconds, st.shade_seconds, ct.rays / ct.paths,
        ct.node_visits / max(1, ct.rays), ct.prim_tests / max(1, ct.rays), ct.node_visits / max(1, 64 * ct.node_wave_steps),
        ct.prim_tests / max(1, 64 * ct.prim_wave_steps)), flush=True)
    dev.close()
