"""Diagnostic: what the tail of a cold 20-pair window consists of.  The same window three ways: complete update() calls; the
same calls with the pose step's result ignored is not possible -- so: (a) update(); (b) compute_3d + detectAndCompute only
(disparity + keypoints of every pair, no matching, no pose); (c) like (b) but only the LAST pair's keypoints are waited for.
usage: python tools/window_parts.py [K]"""
import os, sys, time, gc
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "24")
from openvo_amd import StereoCamera, StereoOdometer
from openvo_amd.synth import Corridor
import bench

K, W = int(sys.argv[1]) if len(sys.argv) > 1 else 20, 5
c = Corridor("C2")
cam = StereoCamera(c.K(), c.dist(), c.K(), c.dist(), c.rect_params(), c.sgbm_params(), (c.w, c.h), max_keypoints=500)
frames = c.pairs(0, W + 4 * K)
staged = cam.stage_pairs(frames)
ctx = cam._ctx
gc.collect(); gc.disable()
odo = StereoOdometer(cam, **bench.ODO_KW)
for i in range(W):
    odo.update(staged[i], None)
for rep in range(4):
    lo = W + (rep % 4) * K
    res = {}
    for mode in ("update", "front", "front_last"):
        odo = StereoOdometer(cam, **bench.ODO_KW)
        odo.update(staged[lo - 1], None)
        odo.reset_lookahead()
        cam.lookahead_stop = lo + K
        ctx.synchronize()
        t0 = time.perf_counter()
        ts = []
        for i in range(lo, lo + K):
            if mode == "update":
                odo.update(staged[i], None)
            else:
                xyz, disp, img = cam.compute_3d(staged[i], None, preprocessed=True)
                if mode == "front" or i == lo + K - 1:
                    kps, desc = odo.orb.detectAndCompute(img, odo.feature_mask(disp))
                    n = len(kps)
            ts.append(time.perf_counter() - t0)
        ctx.synchronize()
        res[mode] = (time.perf_counter() - t0, ts)
        cam.lookahead_stop = None
        odo.reset_lookahead()
    print("rep %d: " % rep + "  ".join("%s %.2f ms (first %.2f, last call %.2f)" % (m, 1e3 * v[0], 1e3 * v[1][0], 1e3 * v[1][-1]) for m, v in res.items()), flush=True)
