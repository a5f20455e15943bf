"""Diagnostic: does importing / initialising torch change the per-frame wall time of the path?"""
import sys, time
sys.path.insert(0, '.')
mode = sys.argv[1] if len(sys.argv) > 1 else "none"
if mode in ("import", "init"):
    import torch
    if mode == "init":
        torch.cuda.is_available(); torch.cuda.synchronize()
import numpy as np
from openvo_amd import StereoCamera, StereoOdometer
from openvo_amd.synth import Corridor
c = Corridor("C2")
cam = StereoCamera(c.K(), c.dist(), c.K(), c.dist(), c.rect_params(), c.sgbm_params(), (c.w, c.h), max_keypoints=500)
odo = StereoOdometer(cam, rigidity_threshold=0.1, outlier_threshold=0.02, preprocessed_frames=True)
frames = [c.pair(i) for i in range(int(sys.argv[2]) if len(sys.argv) > 2 else 40)]
st = cam.stage_pairs(frames)
for i in range(4): odo.update(st[i], None)
cam._ctx.synchronize()
t0 = time.perf_counter()
for i in range(4, len(frames)): odo.update(st[i], None)
cam._ctx.synchronize()
print(mode, "wall ms/frame %.3f" % (1e3 * (time.perf_counter() - t0) / (len(frames) - 4)))
