import os, sys
import numpy as np
sys.path.insert(0, "/root/repo")
from openvo_amd import _native
rng = np.random.default_rng(0)
D=128
p = dict(minDisparity=0, numDisparities=D, blockSize=5, P1=200, P2=800, disp12MaxDiff=1, preFilterCap=63,
         uniquenessRatio=10, speckleWindowSize=100, speckleRange=2)
ctx = _native.Context(0, 2048, 1536, D, 500)
ctx.set_sgbm(p, 0)
for (w, h) in [(1280, 64), (1280, 720)]:
    L = rng.integers(0, 256, (h, w), dtype=np.uint8); R = np.roll(L, -7, axis=1)
    ctx.sgbm_compute_host(L, R)
    print(w, h, "steps total", (w-D+8)//8*8*h, flush=True)
    ctx.sgbm_raster_status()
