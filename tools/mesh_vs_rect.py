import sys, numpy as np
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import raytracing_amd as rt
from conftest import cornell_box_of_meshes
w1, kw = rt.get_scene("cornell_box", spp=4000)
w2, _ = cornell_box_of_meshes()
out = []
for w in (w1, w2):
    d = rt.DeviceScene(w)
    img, _ = d.render(kw["camera"], kw["pattern"], kw["background"], 400, 400)
    d.close()
    out.append(img.astype(np.float64) ** 2)
a, b = out
A = a.reshape(10, 40, 10, 40, 3).mean(axis=(1, 3, 4)); B = b.reshape(10, 40, 10, 40, 3).mean(axis=(1, 3, 4))
np.set_printoptions(precision=3, suppress=True, linewidth=200)
print("rect level\n", A); print("ratio mesh/rect\n", B / np.maximum(A, 1e-9))
print("overall", a.mean(), b.mean(), b.mean() / a.mean())
