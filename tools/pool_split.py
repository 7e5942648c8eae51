#!/usr/bin/env python3
"""Extend / shade time of one frame of cu_cover2 (1000 columns) per slot-pool size (RT_AMD_SLOTS).  python tools/pool_split.py [slots ...]"""
import os
import sys

import torch  # noqa: F401

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import raytracing_amd as rt  # noqa: E402

world, kw = rt.get_scene("cu_cover2", spp=1000)
args = (kw["camera"], kw["pattern"], kw["background"], kw["width"], kw["height"])
for slots in [int(x) for x in sys.argv[1:]] or [4 << 20, 32 << 20]:
    os.environ["RT_AMD_SLOTS"] = str(slots)
    dev = rt.DeviceScene(world)
    dev.render(*args)
    _, st = dev.render(*args, flags=2)
    print("%3d Mi slots: frame %.4f s, extend %.4f s in %d launches (%.1f us), shade %.4f s" % (
        slots >> 20, st.seconds, st.extend_seconds, st.extend_launches, st.extend_seconds / max(1, st.extend_launches) * 1e6, st.shade_seconds), flush=True)
    dev.close()
