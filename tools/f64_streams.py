"""The Float64 frame through rt_render_f64 (null stream) and through accumulate / finalize on a torch stream, with torch
loaded first (its own HIP runtime) -- what bench.py does.  python tools/f64_streams.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import raytracing_amd as rt

cols = 100
world, kw = rt.get_scene("cu_cover2", spp=cols, F=np.float64)
dev = rt.DeviceScene(world, F=np.float64)
W, H = kw["width"], kw["height"]
for _ in range(2):
    img, st = dev.render(kw["camera"], kw["pattern"], kw["background"], W, H)
print("rt_render_f64 (null stream), torch loaded: %.1f Mpaths/s" % (W * H * cols / st.seconds / 1e6), flush=True)
acc = torch.zeros(3 * W * H, dtype=torch.float64, device="cuda")
for name, stream in (("torch current stream", torch.cuda.current_stream().cuda_stream), ("a new torch stream", torch.cuda.Stream().cuda_stream), ("null stream", None)):
    for _ in range(2):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        dev.accumulate(kw["camera"], kw["pattern"], kw["background"], W, H, acc.data_ptr(), stream=stream)
        dev.finalize(acc.data_ptr(), cols, W, H, acc.data_ptr(), stream=stream)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    print("accumulate + finalize on %s: %.1f Mpaths/s" % (name, W * H * cols / dt / 1e6), flush=True)
dev.close()
