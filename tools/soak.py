"""Soak: N pairs through StereoOdometer.run() from host images (a ping-pong walk over 100 rendered frames), then the same through
staged pairs; reports the rate, accepted frames, the sweep-health counter, and the process' resident memory at the start and the
end (a leak in the hand-over paths would show as growth).  usage (gpurun): python tools/soak.py [N]"""
import os, sys, time, gc, resource
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from openvo_amd import StereoCamera, StereoOdometer
from openvo_amd.synth import Corridor
import bench

N = int(sys.argv[1]) if len(sys.argv) > 1 else 40000
c = Corridor("C2")
cam = StereoCamera(c.K(), c.dist(), c.K(), c.dist(), c.rect_params(), c.sgbm_params(), (c.w, c.h), max_keypoints=500)
odo = StereoOdometer(cam, **bench.ODO_KW)
frames = c.pairs(0, 100)
walk = list(range(100)) + list(range(98, 0, -1))


def rss_mb():
    with open("/proc/self/statm") as f:
        return int(f.read().split()[1]) * os.sysconf("SC_PAGE_SIZE") / 1e6


def gen(n):
    for i in range(n):
        yield frames[walk[i % len(walk)]]


for ok in odo.run(gen(200)):
    pass
gc.collect()
r0 = rss_mb()
t0 = time.perf_counter()
acc = 0
for k, ok in enumerate(odo.run(gen(N))):
    acc += bool(ok)
    if k % 10000 == 9999:
        print("  %d pairs, %.0f pairs/s, rss %.0f MB, sweep errors %d" % (k + 1, (k + 1) / (time.perf_counter() - t0), rss_mb(), cam._ctx.sgbm_sweep_status()), flush=True)
cam._ctx.synchronize()
dt = time.perf_counter() - t0
print("from host: %d pairs in %.1f s = %.0f pairs/s, accepted %d, sweep errors %d, rss %.0f -> %.0f MB" % (N, dt, N / dt, acc, cam._ctx.sgbm_sweep_status(), r0, rss_mb()))
