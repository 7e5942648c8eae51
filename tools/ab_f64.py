import os, sys
ROOT = "/root/repo" if os.path.isdir("/root/repo") else os.getcwd()
sys.path.insert(0, ROOT)
import numpy as np
from raytracing_amd import _ffi
_ffi.LIB_PATH = os.path.abspath(sys.argv[1])
import raytracing_amd as rt
for key, cols in (("cu_cover2", 100), ("cornell_box", 100), ("cover1", 64)):
    world, kw = rt.get_scene(key, spp=cols, F=np.float64)
    dev = rt.DeviceScene(world, F=np.float64)
    W, H = kw["width"], kw["height"]
    best = 1e9
    for _ in range(3):
        img, st = dev.render(kw["camera"], kw["pattern"], kw["background"], W, H)
        best = min(best, st.seconds)
    print(sys.argv[1].split("/")[-1], key, "%.1f Mpaths/s" % (W * H * cols / best / 1e6), flush=True)
    dev.close()
