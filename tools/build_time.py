#!/usr/bin/env python3
"""Time of rt_scene_create (flattening on the host, upload, the device trees) with the trees built on the device and,
RT_AMD_HOST_BUILD=1, by the host builder: python tools/build_time.py [scene ...]; a scene "grid:N" is a height field of
2 N^2 triangles."""
import ctypes as C
import os
import sys
import time

import torch  # noqa: F401  (first: its HIP runtime has to be the first in the process)

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import raytracing_amd as rt  # noqa: E402
from raytracing_amd import _ffi as ffi  # noqa: E402
from raytracing_amd.render import SceneManager, build_bvh_scene, flatten_scene  # noqa: E402


def main():
    names = sys.argv[1:] or ["cu_cover2", "cover1", "teapot"]
    L = ffi.lib()
    for name in names:
        if name.startswith("grid:"):
            sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
            from test_gpu_device_build import grid_mesh
            world = [grid_mesh(int(name[5:]))]
        else:
            world = list(rt.get_scene(name, spp=1)[0])
        desc, keep = flatten_scene(world, build_bvh_scene(SceneManager(world)))
        for mode in ("device", "host", "device", "host"):
            if mode == "host":
                os.environ["RT_AMD_HOST_BUILD"] = "1"
            else:
                os.environ.pop("RT_AMD_HOST_BUILD", None)
            handle = C.c_void_p()
            t0 = time.perf_counter()
            ffi.check(L.rt_scene_create(C.byref(desc), 0, C.byref(handle)))
            dt = time.perf_counter() - t0
            info, build = ffi.rtLayout(), ffi.rtBuildInfo()
            ffi.check(L.rt_scene_nodes(handle, C.byref(info), None, 0, None, 0, C.byref(build)))
            print(f"{name:12s} trees by {mode:6s}: rt_scene_create {dt * 1e3:7.2f} ms   (device kernels {build.build_ms:.2f} ms, "
                  f"{info.n_prims} primitives, depth {build.max_depth})", flush=True)
            L.rt_scene_destroy(handle)
        del keep


if __name__ == "__main__":
    main()
