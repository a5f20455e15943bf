"""Timeline of the driver's 20-step window (development aid): when does each update() return, counted from the start of the
timed region?  Same preparation as bench.py (5 warm-up steps, look-ahead reset, everything drained).
Usage: python tools/window20.py [steps] [warmup]"""
import gc, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from openvo_amd import StereoCamera, StereoOdometer
from openvo_amd.synth import Corridor

K = int(sys.argv[1]) if len(sys.argv) > 1 else 20
W = int(sys.argv[2]) if len(sys.argv) > 2 else 5
c = Corridor("C2")
cam = StereoCamera(c.K(), c.dist(), c.K(), c.dist(), c.rect_params(), c.sgbm_params(), (c.w, c.h), max_keypoints=500)
odo = StereoOdometer(cam, rigidity_threshold=0.1, outlier_threshold=0.02, preprocessed_frames=True)
staged = cam.stage_pairs(c.pairs(0, W + K))
ctx = cam._ctx
gc.collect(); gc.disable()
for rep in range(3):
    odo2 = odo if rep == 0 else StereoOdometer(cam, rigidity_threshold=0.1, outlier_threshold=0.02, preprocessed_frames=True)
    for i in range(W):
        odo2.update(staged[i], None)
    cam.reset_lookahead()
    ctx.synchronize()
    t0 = time.perf_counter()
    ts = []
    for i in range(W, W + K):
        odo2.update(staged[i], None)
        ts.append(time.perf_counter() - t0)
    ctx.synchronize()
    dt = time.perf_counter() - t0
    print("window %d: %d steps in %.2f ms (%.0f pairs/s); update() returned at (ms): %s" % (
        rep, K, dt * 1e3, K / dt, " ".join("%.2f" % (t * 1e3) for t in ts)), flush=True)
