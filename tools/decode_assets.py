"""Converts the reference's two asset files into the fixtures this repo ships (run once, in the build container):
  assets/earthmap.jpg -> raytracing_amd/assets/earthmap.rgb8 (+ .json with the shape): one frozen Pillow decode,
                         because the reference's JPEG decoder (Images/FileIO) is unpinned and the GPU box need not
                         have one at all;
  assets/teapot.obj   -> raytracing_amd/assets/teapot.obj (byte copy of the data file).
"""
import json
import os
import shutil
import sys

from PIL import Image
import numpy as np

ref = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "raytracing_amd", "assets")
os.makedirs(out, exist_ok=True)
im = np.asarray(Image.open(os.path.join(ref, "assets", "earthmap.jpg")).convert("RGB"), dtype=np.uint8)
im.tofile(os.path.join(out, "earthmap.rgb8"))
with open(os.path.join(out, "earthmap.json"), "w") as f:
    json.dump({"height": int(im.shape[0]), "width": int(im.shape[1]), "channels": 3,
               "source": "assets/earthmap.jpg decoded with Pillow %s" % Image.__version__}, f)
shutil.copyfile(os.path.join(ref, "assets", "teapot.obj"), os.path.join(out, "teapot.obj"))
print(im.shape)
