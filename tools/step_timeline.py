"""Diagnostic: host-side completion time of every update() in a cold-start window like the driver's (20 steps)."""
import os, sys, time, gc
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "24")
from openvo_amd import StereoCamera, StereoOdometer
from openvo_amd.synth import Corridor
import bench

K, W = int(sys.argv[1]) if len(sys.argv) > 1 else 20, 5
c = Corridor("C2")
cam = StereoCamera(c.K(), c.dist(), c.K(), c.dist(), c.rect_params(), c.sgbm_params(), (c.w, c.h), max_keypoints=500)
odo = StereoOdometer(cam, **bench.ODO_KW)
frames = c.pairs(0, W + K)
staged = cam.stage_pairs(frames)
ctx = cam._ctx
gc.collect(); gc.disable()
for rep in range(3):
    for i in range(W):
        odo.update(staged[i], None)
    cam.reset_lookahead()
    ctx.synchronize()
    t0 = time.perf_counter()
    ts = []
    for i in range(W, W + K):
        odo.update(staged[i], None)
        ts.append(time.perf_counter() - t0)
    ctx.synchronize()
    tend = time.perf_counter() - t0
    print("rep %d: total %.2f ms (%.0f pairs/s); update() returns at ms: %s" % (rep, tend * 1e3, K / tend, " ".join("%.2f" % (t * 1e3) for t in ts)))
    odo = StereoOdometer(cam, **bench.ODO_KW)
