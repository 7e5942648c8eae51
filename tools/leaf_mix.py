#!/usr/bin/env python3
"""How the leaf block of rt_extend_kernel is used (counting variant, RT_AMD_COUNT_MODE): runs and lane tests in all, of
the rectangle / box class and of the sphere class, per 64 rays -- what a leaf stage that runs one class at a time with
every lane busy could save at most.  python tools/leaf_mix.py [scene] [spp]"""
import os
import sys

import torch  # noqa: F401

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import raytracing_amd as rt  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "cu_cover2"
spp = int(sys.argv[2]) if len(sys.argv) > 2 else 64
world, kw = rt.get_scene(name, spp=spp)
dev = rt.DeviceScene(world)
out = {}
for mode, label in ((0, "all"), (1, "rect/box"), (2, "sphere")):
    os.environ["RT_AMD_COUNT_MODE"] = str(mode)
    _, st = dev.render(kw["camera"], kw["pattern"], kw["background"], kw["width"], kw["height"], flags=1)
    out[label] = (st.prim_wave_steps / st.rays * 64, st.prim_tests / st.rays * 64)
    steps, retire = st.node_wave_steps / st.rays * 64, st.bounce_wave_steps / st.rays * 64
dev.close()
print(f"{name} {spp} spp, per 64 rays: step iterations {steps:.2f}, retire/fetch runs {retire:.2f}")
for label, (runs, lanes) in out.items():
    print(f"  leaf block, {label:9s}: runs {runs:.3f}  lane tests {lanes:.1f}  lanes per run {lanes / max(runs, 1e-9):.1f}")
