"""Writes the golden vectors tests/golden/golden_<scene>.npz from the CPU oracle: closest hits of a fixed 32x32 primary
ray set, the radiance of 256 fixed paths and a 24x16 image at 4 pattern columns, per scene.  They freeze the oracle's
behaviour (so a change to the oracle shows up as a diff here) and are what the GPU path is compared with on the GPU box,
where the reference tree does not exist.  The reference itself has no tests or vectors to take these from."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, os.path.dirname(HERE))
import raytracing_amd as rt  # noqa: E402
import oracle_ffi as O  # noqa: E402

SCENES = ["cornell_box", "cornell_smoke_box", "cover1", "cu_cover2", "cover2", "teapot"]
W, H, SPP = 24, 16, 4


def golden_inputs(name):
    world, kw = rt.get_scene(name, spp=SPP)
    cam, pat, bg = kw["camera"], kw["pattern"], kw["background"]
    rays = O.primary_rays(cam, W, H, 32, 32)
    ids = np.array([(i, j, n) for j in range(0, H, 2) for i in range(0, W, 3) for n in range(SPP)], np.int32)
    return world, cam, pat, bg, rays, ids


if __name__ == "__main__":
    for name in SCENES:
        world, cam, pat, bg, rays, ids = golden_inputs(name)
        ref = O.OracleScene(world)
        hits = ref.trace_rays(rays)
        rad = ref.trace_paths(cam, pat, bg, W, H, ids)
        img = ref.render(cam, pat, bg, W, H)
        np.savez_compressed(os.path.join(HERE, f"golden_{name}.npz"), rays=rays, ids=ids, radiance=rad, image=img,
                            **{"hit_" + k: v for k, v in hits.items()})
        print(name, "hits", int((hits["prim_id"] >= 0).sum()), "image mean", img.mean())
