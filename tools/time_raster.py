"""Timing aid: SGBM stage times alone on the GPU for several image geometries (separates the per-column step
time of the raster sweep from the per-band hand-off lag)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from openvo_amd import _native
rng = np.random.default_rng(0)
D = int(os.environ.get("TR_D", "128"))
mode = int(os.environ.get("TR_MODE", "0"))
p = dict(minDisparity=0, numDisparities=D, blockSize=5, P1=200, P2=800, disp12MaxDiff=1, preFilterCap=63,
         uniquenessRatio=10, speckleWindowSize=100, speckleRange=2)
ctx = _native.Context(0, 2048, 1536, D, 500)
ctx.set_sgbm(p, mode)
for (w, h) in [(1280, 16), (1280, 24), (1280, 32), (1280, 64), (1280, 128), (1280, 720), (D + 256, 16), (D + 256, 32), (D + 256, 64), (D + 256, 720)]:
    L = rng.integers(0, 256, (h, w), dtype=np.uint8)
    R = np.roll(L, -7, axis=1)
    ctx.sgbm_compute_host(L, R)
    ctx.enable_timing(True); ctx.timings(reset=True)
    n = 5
    for _ in range(n):
        ctx.sgbm_compute_host(L, R)
    tm = ctx.timings(reset=True); ctx.enable_timing(False)
    print("%4dx%-4d W1=%4d bands=%3d  " % (w, h, w - D, (h + 7) // 8) + "  ".join("%s=%.3f" % (k, v[0] / n) for k, v in tm.items() if k.startswith("sgbm")), flush=True)
print("status", ctx.sgbm_raster_status())
