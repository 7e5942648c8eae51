"""One Float64 frame of a sample scene, for profiling: python tools/f64_one.py [scene] [columns]"""
import sys

import numpy as np

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import raytracing_amd as rt

key = sys.argv[1] if len(sys.argv) > 1 else "cu_cover2"
cols = int(sys.argv[2]) if len(sys.argv) > 2 else 64
world, kw = rt.get_scene(key, spp=cols, F=np.float64)
dev = rt.DeviceScene(world, F=np.float64)
W, H = kw["width"], kw["height"]
for _ in range(2):
    img, st = dev.render(kw["camera"], kw["pattern"], kw["background"], W, H)
print("%s float64 %d columns: %.1f Mpaths/s, %.3f s" % (key, cols, W * H * cols / st.seconds / 1e6, st.seconds))
dev.close()
