#!/usr/bin/env python3
"""Two renders of half the pattern columns each, on one GPU at the same time (two scenes, two host threads, the library's
own streams), against one render of all the columns: what overlapping one pool's extend kernel with the other's shade kernel
could win.  python tools/two_renders.py [columns] [slots per scene in Mi]"""
import os
import sys
import threading
import time

import torch  # noqa: F401

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import raytracing_amd as rt  # noqa: E402

cols = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
if len(sys.argv) > 2:
    os.environ["RT_AMD_SLOTS"] = str(int(sys.argv[2]) << 20)
world, kw = rt.get_scene("cu_cover2", spp=cols)
W, H = kw["width"], kw["height"]
one = rt.DeviceScene(world)
one.render(kw["camera"], kw["pattern"], kw["background"], W, H)
t0 = time.perf_counter()
one.render(kw["camera"], kw["pattern"], kw["background"], W, H)
single = time.perf_counter() - t0
one.close()
halves = [rt.DeviceScene(world) for _ in range(2)]
pats = [kw["pattern"][:, k::2].copy() for k in range(2)]


def work(k):
    halves[k].render(kw["camera"], pats[k], kw["background"], W, H)


for rep in range(2):
    th = [threading.Thread(target=work, args=(k,)) for k in range(2)]
    t0 = time.perf_counter()
    for t in th:
        t.start()
    for t in th:
        t.join()
    both = time.perf_counter() - t0
for h in halves:
    h.close()
paths = W * H * cols
print("%d columns: one render %.1f ms (%.1f Mpaths/s), two at once %.1f ms (%.1f Mpaths/s)" % (
    cols, single * 1e3, paths / single / 1e6, both * 1e3, paths / both / 1e6))
