"""Experiment: do two independent renders that share one GPU overlap well?  Starts N processes that render the same
scene with spp/N each at the same moment and reports wall time from the common start to the last finish."""
import os, sys, time, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    sys.path.insert(0, ROOT)
    import raytracing_amd as rt
    name, spp, t_start = sys.argv[2], int(sys.argv[3]), float(sys.argv[4])
    world, kw = rt.get_scene(name, spp=spp)
    dev = rt.DeviceScene(world)
    args = (kw["camera"], kw["pattern"], kw["background"], kw["width"], kw["height"])
    dev.render(*args)
    while time.time() < t_start:
        pass
    t0 = time.time()
    for _ in range(3):
        img, st = dev.render(*args)
    t1 = time.time()
    print("child spp %d: start +%.3f end +%.3f  (%.3f s per frame, kernel %.3f)" % (spp, t0 - t_start, t1 - t_start, (t1 - t0) / 3, st.seconds), flush=True)
else:
    name, spp, n = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
    t_start = time.time() + 25.0
    procs = [subprocess.Popen([sys.executable, "-u", __file__, "child", name, str(spp // n), str(t_start)]) for _ in range(n)]
    rc = [p.wait() for p in procs]
    print("n=%d done rc=%s" % (n, rc))
