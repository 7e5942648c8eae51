"""Statistics of the reference's gallery images (outputs/*.png), the only artefacts of the reference that pin converged
results.  Run in the build container (needs /root/reference and Pillow); the files it writes are the fixtures, the
PNGs themselves never leave the reference tree.

gallery_stats.json -- the two deterministic scenes (no rand() in their construction): linear-space mean RGB (the PNG
stores clamp01(sqrt(c)), so linear = (v/255)^2), the width of the black frame, and the grid of block means used for a
block-RMSE comparison of the whole picture.
gallery_blocks.npz -- the scenes that call the unseeded rand() while they are built (cover1, cover2, cu_cover2): block
means in gamma space (8 x 8 pixels, float32).  Only the parts of those pictures that do not depend on the random
part of the scene can be compared (sky, the three large spheres, the light, the moving / glass / metal / medium / earth
spheres); the test decides which blocks those are by rendering the scene with several scene seeds.
"""
import json
import os
import sys

import numpy as np
from PIL import Image

ref = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
here = os.path.dirname(os.path.abspath(__file__))
out = {}
for name, fn, block in [("cornell_box_1000", "cornell_box_1000_samples.png", 16),
                        ("cornell_box_10000", "cornell_box_10000_samples.png", 16),
                        ("cornell_smoke_box_10000", "cornell_smoke_box_10000_samples.png", 24)]:
    im = np.asarray(Image.open(os.path.join(ref, "outputs", fn)).convert("RGB"), dtype=np.float64) / 255.0
    lin = im ** 2
    h, w, _ = lin.shape
    gh, gw = h // block, w // block
    blocks = lin[:gh * block, :gw * block].reshape(gh, block, gw, block, 3).mean(axis=(1, 3))
    gblocks = im[:gh * block, :gw * block].reshape(gh, block, gw, block, 3).mean(axis=(1, 3))
    row = im[h // 2].sum(axis=1)
    frame = int(np.argmax(row > 0.02))
    out[name] = {"width": w, "height": h, "block": block, "linear_mean": lin.mean(axis=(0, 1)).tolist(),
                 "gamma_mean": im.mean(axis=(0, 1)).tolist(), "black_frame_px": frame,
                 "block_means_linear": np.round(blocks, 5).tolist(), "block_means_gamma": np.round(gblocks, 5).tolist()}
    print(name, out[name]["linear_mean"], "frame", frame)
with open(os.path.join(here, "gallery_stats.json"), "w") as f:
    json.dump(out, f)

BLOCK = 8
arrays = {}
for name, fn in [("cover1_500", "cover1_500_samples.png"), ("cu_cover2_1000", "cu_cover2_1000_samples.png"),
                 ("cover2_10000", "cover2_10000_samples.png"), ("cover2_100", "cover2_100_samples.png")]:
    im = np.asarray(Image.open(os.path.join(ref, "outputs", fn)).convert("RGB"), dtype=np.float64) / 255.0
    h, w, _ = im.shape
    gh, gw = h // BLOCK, w // BLOCK
    arrays[name] = im[:gh * BLOCK, :gw * BLOCK].reshape(gh, BLOCK, gw, BLOCK, 3).mean(axis=(1, 3)).astype(np.float32)
    print(name, arrays[name].shape, arrays[name].mean(axis=(0, 1)))
np.savez_compressed(os.path.join(here, "gallery_blocks.npz"), block=np.int32(BLOCK), **arrays)
