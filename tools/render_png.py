"""Renders a configuration the way src/main.cpp shows it (running mean over frames of 100 samples, linear values
clamped to 0..1 and written as 8 bits, as the blit of main.cpp:152 into the window does) and saves a PNG.
usage: python tools/render_png.py <config> <width> <height> <frames> <out.png> [mesh.off]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from PIL import Image
import ray_tracing_amd as rta

name, w, h, frames, out = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), sys.argv[5]
kw = {"mesh": sys.argv[6]} if len(sys.argv) > 6 else {}
cfg = rta.scenes.CONFIGS[name](width=w, height=h, spp=100, **kw)
rt = rta.scenes.make_renderer(cfg)
for n in range(1, frames + 1):
    rt.compute(w, h)                      # main.cpp:145
    rt.convert(n)                         # main.cpp:147-151
img = rt.readImage()
rt.close()
rgb = (np.clip(img[..., :3], 0.0, 1.0) * 255.0 + 0.5).astype(np.uint8)
# the texture's row 0 is the bottom of the window (GL); the converter already flipped the rows for that
Image.fromarray(rgb[::-1]).save(out)
print(out, rgb.shape, float(img[..., :3].mean()))
