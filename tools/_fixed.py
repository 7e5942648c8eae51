import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import raytracing_amd as rt
world, kw = rt.get_scene("cu_cover2", spp=1000)
dev = rt.DeviceScene(world)
W, H = kw["width"], kw["height"]
acc = torch.zeros(3 * W * H, dtype=torch.float32, device="cuda")
stream = torch.cuda.current_stream().cuda_stream
res = {}
for spp in (1000, 125, 250, 500, 1000, 125):
    pat = rt.Halton_sequence32(spp)
    best = 1e9
    for _ in range(4):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        dev.accumulate(kw["camera"], pat, kw["background"], W, H, acc.data_ptr(), stream=stream)
        torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
    res[spp] = best
    print(spp, "spp: %.2f ms, %.1f Mpaths/s" % (best * 1e3, W * H * spp / best / 1e6), flush=True)
per_col = (res[1000] - res[125]) / 875
print("per column %.4f ms, fixed %.2f ms" % (per_col * 1e3, (res[125] - 125 * per_col) * 1e3))
