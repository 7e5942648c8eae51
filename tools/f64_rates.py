"""Rates of the Float64 mode (rt_render_f64) on the BASELINE configurations at a reduced number of pattern columns,
beside the Float32 path on the same frame.  usage: python tools/f64_rates.py [columns]"""
import sys
import time

import numpy as np

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import raytracing_amd as rt

cols = int(sys.argv[1]) if len(sys.argv) > 1 else 16
for key in ("cu_cover2", "cover1", "teapot", "cornell_box", "cornell_smoke_box"):
    line = [key]
    for F in (np.float64, np.float32):
        world, kw = rt.get_scene(key, spp=cols, F=F)
        dev = rt.DeviceScene(world, F=F)
        W, H = kw["width"], kw["height"]
        dev.render(kw["camera"], kw["pattern"][:, :2], kw["background"], W, H)
        t0 = time.perf_counter()
        img, st = dev.render(kw["camera"], kw["pattern"], kw["background"], W, H)
        dt = time.perf_counter() - t0
        line.append("%s %.1f Mpaths/s (%.3f s device, %.3f s wall)" % (np.dtype(F).name, W * H * cols / st.seconds / 1e6, st.seconds, dt))
        dev.close()
    print("  ".join(line), flush=True)
