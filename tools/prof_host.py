import sys, time
sys.path.insert(0, '.')
import numpy as np
from openvo_amd import StereoCamera, StereoOdometer
from openvo_amd.synth import Corridor
c = Corridor("C2")
cam = StereoCamera(c.K(), c.dist(), c.K(), c.dist(), c.rect_params(), c.sgbm_params(), (c.w, c.h), max_keypoints=500)
odo = StereoOdometer(cam, rigidity_threshold=0.1, outlier_threshold=0.02, preprocessed_frames=True)
frames = [c.pair(i) for i in range(24)]
st = cam.stage_pairs(frames)
ctx = cam._ctx
T = {}
def wrap(obj, name, label=None):
    f = getattr(obj, name)
    label = label or name
    def g(*a, **k):
        t0 = time.perf_counter(); r = f(*a, **k); T.setdefault(label, []).append(time.perf_counter() - t0); return r
    setattr(obj, name, g)
for n in ("load_staged_pair", "sgbm_compute", "orb_slot", "point_clouds", "rigid_clique", "umeyama", "synchronize"):
    wrap(ctx, n)
wrap(cam, "compute_3d"); wrap(odo, "point_cloud_transform"); wrap(odo, "point_clouds", "odo.point_clouds")
wrap(odo.orb, "detectAndCompute")
for i in range(4): odo.update(st[i], None)
T.clear()
t0 = time.perf_counter()
for i in range(4, 24):
    odo.update(st[i], None)
ctx.synchronize()
tot = time.perf_counter() - t0
print("per frame ms: %.3f" % (1e3 * tot / 20))
for k, v in T.items():
    print("%-24s n/frame=%.1f  ms/frame=%.3f" % (k, len(v) / 20, 1e3 * sum(v) / 20))
# pure kernel time without host: enqueue sgbm only
t0 = time.perf_counter()
for i in range(4, 24):
    ctx._lib.vo_load_staged_pair(ctx._h, 0, i, 1); ctx._lib.vo_sgbm_compute(ctx._h, 0, None)
ctx.synchronize()
print("sgbm only per frame ms: %.3f" % (1e3 * (time.perf_counter() - t0) / 20))
