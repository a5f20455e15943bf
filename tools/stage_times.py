"""Per-stage GPU time (hipEvent) of the C2 path, for tuning runs: python tools/stage_times.py"""
import sys, time
sys.path.insert(0, '.')
import numpy as np
from openvo_amd import StereoCamera, StereoOdometer
from openvo_amd.synth import Corridor
c = Corridor(sys.argv[1] if len(sys.argv) > 1 else "C2")
cam = StereoCamera(c.K(), c.dist(), c.K(), c.dist(), c.rect_params(), c.sgbm_params(), (c.w, c.h), max_keypoints=500)
odo = StereoOdometer(cam, rigidity_threshold=0.1, outlier_threshold=0.02, preprocessed_frames=True)
frames = [c.pair(i) for i in range(16)]
st = cam.stage_pairs(frames)
ctx = cam._ctx
for i in range(4): odo.update(st[i], None)
ctx.enable_timing(True); ctx.timings(reset=True)
for i in range(4, 16): odo.update(st[i], None)
tm = ctx.timings(reset=True); ctx.enable_timing(False)
print(" ".join("%s=%.3f" % (k, v[0] / 12) for k, v in tm.items()), "sum=%.3f" % (sum(v[0] for v in tm.values()) / 12))
t0 = time.perf_counter()
for r in range(3):
    for i in range(4, 16): odo.update(st[i], None)
ctx.synchronize()
print("wall ms/frame %.3f" % (1e3 * (time.perf_counter() - t0) / 36))
