#!/usr/bin/env python3
"""A height field of 2 N^2 triangles seen from above and from inside a valley: Mpaths/s with the walk over pair nodes
and (RT_AMD_NO_PAIRS=1) with the stackless walk, and the depth of the device tree.  python tools/deep_tree_rates.py [N] [spp]"""
import os
import sys

import torch  # noqa: F401

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import raytracing_amd as rt  # noqa: E402
from raytracing_amd.hittables import Camera, Vec3  # noqa: E402
from test_gpu_device_build import grid_mesh  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 362
spp = int(sys.argv[2]) if len(sys.argv) > 2 else 64
world = [grid_mesh(N)]
pos = world[0].vertices[:, :3]
lo, hi = pos.min(axis=0), pos.max(axis=0)
mid, ext = (lo + hi) / 2, float((hi - lo).max())
origin = Vec3(mid[0] + 0.3 * ext, hi[1] + 0.6 * ext, mid[2] - 0.9 * ext)
cam = Camera(origin, origin - Vec3(mid[0], mid[1], mid[2]), Vec3(0, 1, 0), 50.0, 1.0, 0.0, 10.0, 0.0, 1.0)
pat, bg, W, H = rt.Halton_sequence32(spp), Vec3(0.7, 0.8, 1.0), 800, 800
imgs = {}
for mode in ("pairs", "stackless"):
    if mode == "stackless":
        os.environ["RT_AMD_NO_PAIRS"] = "1"
    else:
        os.environ.pop("RT_AMD_NO_PAIRS", None)
    dev = rt.DeviceScene(world)
    build = dev.layout()[3]
    dev.render(cam, pat, bg, W, H)
    img, st = dev.render(cam, pat, bg, W, H, flags=2)
    imgs[mode] = img
    print(f"grid {N}: {2 * N * N} triangles, depth {build.max_depth}, {mode:9s}: {W * H * spp / st.seconds / 1e6:8.1f} Mpaths/s "
          f"(extend {st.extend_seconds * 1e3:.1f} ms, shade {st.shade_seconds * 1e3:.1f} ms)", flush=True)
    dev.close()
print("pictures identical:", bool((imgs["pairs"].view("uint32") == imgs["stackless"].view("uint32")).all()))
