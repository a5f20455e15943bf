"""Diagnostic: why do the first cold windows of bench.py read low?  Runs R cold 20-pair windows after W warm-up steps (like the
driver's command) and prints, per window, its rate, the shader clock (vo_shader_clock) and the power-management levels the
kernel driver reports in sysfs (sclk / mclk / fclk / socclk: the starred entry of pp_dpm_*), then the same again after a
second warm-up of 400 pairs.  usage: python tools/window_drift.py [W] [R]"""
import glob, os, sys, time, gc
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "24")
from openvo_amd import StereoCamera, StereoOdometer
from openvo_amd.synth import Corridor
import bench

W = int(sys.argv[1]) if len(sys.argv) > 1 else 5
R = int(sys.argv[2]) if len(sys.argv) > 2 else 8
K = 20


def dpm():
    out = {}
    for name in ("sclk", "mclk", "fclk", "socclk"):
        for f in glob.glob("/sys/class/drm/card*/device/pp_dpm_%s" % name):
            try:
                star = [l.strip() for l in open(f) if "*" in l]
                out[name] = star[0] if star else "?"
            except Exception as e:
                out[name] = "n/a"
            break
    return out


c = Corridor("C2")
cam = StereoCamera(c.K(), c.dist(), c.K(), c.dist(), c.rect_params(), c.sgbm_params(), (c.w, c.h), max_keypoints=500)
frames = c.pairs(0, 105)
staged = cam.stage_pairs(frames)
ctx = cam._ctx
gc.collect(); gc.disable()
print("idle:", dpm(), flush=True)


def windows(tag, first):
    odo = StereoOdometer(cam, **bench.ODO_KW)
    for i in range(W):
        odo.update(staged[i], None)
    for r in range(R):
        lo = first + (r * K) % 80
        odo.reset_lookahead()
        cam.lookahead_stop = lo + K
        ctx.synchronize()
        t0 = time.perf_counter()
        for i in range(lo, lo + K):
            odo.update(staged[i], None)
        ctx.synchronize()
        dt = time.perf_counter() - t0
        print("%s window %d: %.0f pairs/s  shader %.0f MHz  %s" % (tag, r, K / dt, ctx.shader_clock(200), dpm()), flush=True)
    cam.lookahead_stop = None
    odo.reset_lookahead()


windows("cold ", W)
odo = StereoOdometer(cam, **bench.ODO_KW)
t0 = time.perf_counter()
for rep in range(5):
    for i in range(80):
        odo.update(staged[i], None)
ctx.synchronize()
print("400 pairs in %.0f ms" % (1e3 * (time.perf_counter() - t0)), flush=True)
odo.reset_lookahead()
windows("warm ", W)
time.sleep(2.0)
print("after 2 s idle:", dpm(), flush=True)
windows("rest ", W)
