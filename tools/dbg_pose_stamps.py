"""Diagnostic: phase times inside k_pose_solve (needs the library built with -DVO_POSE_STAMPS as openvo_amd/libvo355_dbg.so)."""
import ctypes, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from openvo_amd import _native
_native.LIB_PATH = os.path.join(os.path.dirname(_native.LIB_PATH), "libvo355_dbg.so")
from openvo_amd import StereoCamera, StereoOdometer
from openvo_amd.synth import Corridor
import bench

c = Corridor("C2")
cam = StereoCamera(c.K(), c.dist(), c.K(), c.dist(), c.rect_params(), c.sgbm_params(), (c.w, c.h), max_keypoints=500)
odo = StereoOdometer(cam, **bench.ODO_KW)
L = _native.lib()
L.vo_debug_pose_stamps.argtypes = [ctypes.c_void_p]
names = {(0, 10): "load bits", (10, 11): "clique loop", (11, 1): "compaction", (1, 2): "sync", (2, 3): "umeyama 1", (3, 4): "residuals",
         (4, 5): "median", (5, 6): "outlier compaction", (6, 9): "umeyama 2", (6, 12): "u2 sums1", (12, 13): "u2 sums2", (13, 14): "u2 svd", (14, 9): "u2 rest"}
for k in range(6):
    Lk, Rk = c.pair(k)
    odo.update(Lk, Rk)
    cam._ctx.synchronize()
    st = np.zeros(32, np.int64)
    L.vo_debug_pose_stamps(st.ctypes.data)
    if k == 0:
        continue
    line = ["pair %d: m=%d clique=%d total %.1f us |" % (k, st[21], st[20], (st[9] - st[0]) / 100.0)]
    for (a, b), nm in names.items():
        line.append("%s %.1f" % (nm, (st[b] - st[a]) / 100.0))
    print("  ".join(line))
