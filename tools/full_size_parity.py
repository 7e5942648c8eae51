"""Full-size frames of every sample scene against the oracle, bit for bit (a few pattern columns each; the oracle takes
seconds per scene on 16 cores).  Not part of the test suite because of its run time; run on the GPU box:
    python tools/full_size_parity.py [spp scale] [f64]      # f64: the Float64 mode against the oracle built for Float64"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, raytracing_amd as rt, oracle_ffi as O
scale = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
F = np.float64 if "f64" in sys.argv[2:] else np.float32
U = np.uint64 if F is np.float64 else np.uint32
bad = 0
for name, spp in (("cu_cover2", 48), ("cornell_box", 64), ("cornell_smoke_box", 64), ("cover2", 16), ("cover1", 8), ("teapot", 4)):
    spp = max(1, int(spp * scale))
    world, kw = rt.get_scene(name, spp=spp, F=F)
    ref, dev = O.OracleScene(world, F=F), rt.DeviceScene(world, F=F)
    W, H = kw["width"], kw["height"]
    t0 = time.time()
    a = ref.render(kw["camera"], kw["pattern"], kw["background"], W, H, threads=16)
    b, st = dev.render(kw["camera"], kw["pattern"], kw["background"], W, H)
    neq = (a.view(U) != b.view(U)).any(axis=2)
    bad += int(neq.sum())
    print(name, spp, "spp,", W * H * spp, "paths: pixels differing", int(neq.sum()), "max abs", float(np.abs(a - b).max()),
          "(%.1f s)" % (time.time() - t0), flush=True)
    dev.close()
sys.exit(1 if bad else 0)
