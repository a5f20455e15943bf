"""cProfile of the bench loop's host side (where the Python interpreter and the native calls spend a frame)."""
import cProfile, gc, os, pstats, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from openvo_amd import StereoCamera, StereoOdometer, sharding
from openvo_amd.synth import Corridor
c = Corridor("C2")
cam = StereoCamera(c.K(), c.dist(), c.K(), c.dist(), c.rect_params(), c.sgbm_params(), (c.w, c.h), max_keypoints=500)
odo = StereoOdometer(cam, rigidity_threshold=0.1, outlier_threshold=0.02, preprocessed_frames=True)
N = 260
st = cam.stage_pairs(c.pairs(0, N))
for i in range(20): odo.update(st[i], None)
gc.collect(); gc.disable()
def loop():
    for i in range(20, N):
        before = odo.c_T_w
        ok = odo.update(st[i], None)
        sharding.relative_from_chain(before, odo.c_T_w)
pr = cProfile.Profile(); pr.enable(); loop(); pr.disable()
ps = pstats.Stats(pr); ps.sort_stats("tottime").print_stats(22)
