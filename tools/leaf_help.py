#!/usr/bin/env python3
"""Per leaf-block run of rt_extend_kernel (counting variant, RT_AMD_COUNT_MODE=3): lanes that hold no ray, lanes whose
second waiting leaf is not a box, and how many of the latter the former could take.  python tools/leaf_help.py [scene] [spp]"""
import os
import sys

import torch  # noqa: F401

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import raytracing_amd as rt  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "cu_cover2"
spp = int(sys.argv[2]) if len(sys.argv) > 2 else 64
world, kw = rt.get_scene(name, spp=spp)
dev = rt.DeviceScene(world)
os.environ["RT_AMD_COUNT_MODE"] = "3"
_, st = dev.render(kw["camera"], kw["pattern"], kw["background"], kw["width"], kw["height"], flags=1)
dev.close()
runs = max(1, st.prim_wave_steps)
print(f"{name} {spp} spp: leaf-block runs per 64 rays {runs / st.rays * 64:.2f}; per run: idle lanes {st.prim_tests / runs:.1f}, "
      f"second leaves {st.node_visits / runs:.1f}, could be taken {st.node_wave_steps / runs:.1f}")

# the same for the step loop (RT_AMD_COUNT_MODE=4): lanes without a ray per iteration, while the wave can still fetch slots
# and after the launch has run out of them
dev = rt.DeviceScene(world)
os.environ["RT_AMD_COUNT_MODE"] = "4"
_, st = dev.render(kw["camera"], kw["pattern"], kw["background"], kw["width"], kw["height"], flags=1)
dev.close()
a, b = max(1, st.node_wave_steps), max(1, st.prim_wave_steps)
print(f"step iterations per 64 rays: {a / st.rays * 64:.2f} with slots left to fetch ({st.prim_tests / a:.1f} lanes without a ray), "
      f"{b / st.rays * 64:.2f} after the slots ran out ({st.node_visits / b:.1f} lanes without a ray)")
