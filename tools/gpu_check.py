"""Quick GPU-vs-oracle comparison over all scenes (development aid; the real checks live in tests/)."""
import sys, time, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import raytracing_amd as rt
import oracle_ffi as O

names = sys.argv[1:] or ["cornell_box", "cornell_smoke_box", "cu_cover2", "cover1", "cover2", "teapot"]
for name in names:
    world, kw = rt.get_scene(name, spp=8)
    cam, pat, bg = kw["camera"], kw["pattern"], kw["background"]
    W, H = 96, 64
    ref = O.OracleScene(world)
    dev = rt.DeviceScene(world)
    rays = O.primary_rays(cam, W, H, 96, 96)
    a = ref.trace_rays(rays); b = dev.trace_rays(rays)
    hit = a["prim_id"] >= 0
    print(name, "rays: prim mismatches", int((a["prim_id"] != b["prim_id"]).sum()), "sub", int((a["sub_id"] != b["sub_id"]).sum()),
          "t bit-mismatch", int((a["t"].view(np.uint32) != b["t"].view(np.uint32)).sum()),
          "normal maxdiff", float(np.abs(a["normal"] - b["normal"]).max()), "uv maxdiff", float(np.abs(a["uv"] - b["uv"]).max()),
          "front", int((a["front"] != b["front"]).sum()), "hits", int(hit.sum()))
    ids = np.array([(i, j, n) for j in range(0, H, 2) for i in range(0, W, 2) for n in range(4)], np.int32)
    ra = ref.trace_paths(cam, pat, bg, W, H, ids); rb = dev.trace_paths(cam, pat, bg, W, H, ids)
    neq = (ra.view(np.uint32) != rb.view(np.uint32)).any(axis=1)
    print(name, "paths:", len(ids), "bit-mismatching paths", int(neq.sum()), "max abs diff", float(np.nanmax(np.abs(ra - rb))))
    t0 = time.time(); ia = ref.render(cam, pat, bg, W, H); t1 = time.time()
    ib, st = dev.render(cam, pat, bg, W, H); t2 = time.time()
    d = np.abs(ia - ib)
    print(name, "image: max diff", float(d.max()), "n>1e-5", int((d > 1e-5).sum()), "cpu %.2fs gpu %.3fs (kernel %.4fs)" % (t1 - t0, t2 - t1, st.seconds))
