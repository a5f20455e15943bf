"""Per-stage GPU elapsed time per pair while the look-ahead engines overlap (events on every stage: the
run is perturbed by the extra event packets, the split is indicative), next to the same with look-ahead off."""
import gc, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from openvo_amd import StereoCamera, StereoOdometer
from openvo_amd.synth import Corridor

c = Corridor("C2")
cam = StereoCamera(c.K(), c.dist(), c.K(), c.dist(), c.rect_params(), c.sgbm_params(), (c.w, c.h), max_keypoints=500)
N = 140
st = cam.stage_pairs(c.pairs(0, N))
ctx = cam._ctx
for la in (cam.lookahead, 0):
    cam.reset_lookahead(); cam.lookahead = la
    odo = StereoOdometer(cam, rigidity_threshold=0.1, outlier_threshold=0.02, preprocessed_frames=True)
    for i in range(20): odo.update(st[i], None)
    ctx.synchronize(); ctx.timings(reset=True); ctx.enable_timing(True)
    gc.collect(); gc.disable()
    t0 = time.perf_counter()
    for i in range(20, N): odo.update(st[i], None)
    ctx.synchronize(); dt = time.perf_counter() - t0
    gc.enable()
    tm = ctx.timings(reset=True); ctx.enable_timing(False)
    n = N - 20
    print("lookahead %d: %.3f ms/pair wall; stage ms/pair: %s ; sum %.3f" % (
        la, 1e3 * dt / n, {k: round(v[0] / n, 3) for k, v in tm.items()}, sum(v[0] for v in tm.values()) / n))
