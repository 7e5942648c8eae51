#!/usr/bin/env python3
"""Walk statistics of rt_extend_kernel for one frame (RT_FLAG_COUNTERS): box tests and primitive tests per ray, and how
full the waves are in the step loop and in the leaf block: python tools/walk_stats.py [scene] [spp]"""
import os
import sys

import torch  # noqa: F401  (first: its HIP runtime has to be the first in the process)

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import raytracing_amd as rt  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "cu_cover2"
spp = int(sys.argv[2]) if len(sys.argv) > 2 else 64
world, kw = rt.get_scene(name, spp=spp)
dev = rt.DeviceScene(world)
img, st = dev.render(kw["camera"], kw["pattern"], kw["background"], kw["width"], kw["height"], flags=1)  # RT_FLAG_COUNTERS
used = dev.pairs()[1]
dev.close()
print(f"{name} {spp} spp, pair nodes {'walked' if used else 'not walked'}: rays/path {st.rays / st.paths:.2f}  box tests/ray "
      f"{st.node_visits / st.rays:.2f}  prim tests/ray {st.prim_tests / st.rays:.2f}  step-block wave iterations/ray "
      f"{st.node_wave_steps / st.rays * 64:.2f} x 1/64  leaf-block runs/ray {st.prim_wave_steps / st.rays * 64:.2f} x 1/64  "
      f"retire/fetch runs/ray {st.bounce_wave_steps / st.rays * 64:.2f} x 1/64")
