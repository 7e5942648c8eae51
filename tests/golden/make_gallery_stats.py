"""Statistics of the reference's deterministic gallery images (outputs/*.png), the only artefacts of the
reference that pin converged results.  Run in the build container (needs /root/reference and Pillow); the JSON it
writes is the fixture, the PNGs themselves never leave the reference tree.

For each image: linear-space mean RGB (the PNG stores clamp01(sqrt(c)), so linear = (v/255)^2), the width of the
black frame, and the grid of block means (linear space) used for a block-RMSE comparison.
"""
import json
import os
import sys

import numpy as np
from PIL import Image

ref = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
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
with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "gallery_stats.json"), "w") as f:
    json.dump(out, f)
