"""Where the host thread waits in the steady state of the HBM-resident loop (bench.py's `steady_state` shape): wall time of each
native call per pair -- which calls wait for the GPU (a frame's disparity + keypoints, a pose step), which only enqueue."""
import gc, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from openvo_amd import StereoCamera, StereoOdometer
from openvo_amd.synth import Corridor
import bench

N = int(os.environ.get("HB_FRAMES", "105"))
STEPS = int(os.environ.get("HB_STEPS", "480"))
c = Corridor("C2")
cam = StereoCamera(c.K(), c.dist(), c.K(), c.dist(), c.rect_params(), c.sgbm_params(), (c.w, c.h), max_keypoints=500)
odo = StereoOdometer(cam, **bench.ODO_KW)
frames = c.pairs(0, N)
seq = []
while len(seq) < STEPS + 20:                                   # forwards and backwards: consecutive frames either way
    seq += list(range(N)) + list(range(N - 2, 0, -1))
seq = seq[:STEPS + 20]
staged = cam.stage_pairs([frames[i] for i in seq])
ctx = cam._ctx
T = {}
def wrap(obj, name):
    f = getattr(obj, name)
    def g(*a, **k):
        t0 = time.perf_counter(); r = f(*a, **k); T.setdefault(name, []).append(time.perf_counter() - t0); return r
    setattr(obj, name, g)
for n in dir(ctx):
    if not n.startswith("_") and callable(getattr(ctx, n)) and n not in ("close",):
        wrap(ctx, n)
for j in range(20):
    odo.update(staged[j], None)
from openvo_amd import _native as _nat
_lib = _nat.lib()
if os.environ.get("VO_POSE_TRACE") and _lib is not None:
    _lib.vo_debug_pose_trace_dump()
T.clear()
gc.collect(); gc.disable()
t0 = time.perf_counter()
for j in range(20, 20 + STEPS):
    odo.update(staged[j], None)
ctx.synchronize()
tot = time.perf_counter() - t0
n = STEPS
print("per pair ms: %.3f  (%.1f pairs/s)" % (1e3 * tot / n, n / tot))
acc = 0.0
for k, v in sorted(T.items(), key=lambda kv: -sum(kv[1])):
    acc += sum(v)
    print("%-26s calls/pair=%.2f  ms/pair=%.4f  us/call=%.1f" % (k, len(v) / n, 1e3 * sum(v) / n, 1e6 * sum(v) / len(v)))
print("native calls total ms/pair: %.3f ; python outside native: %.3f" % (1e3 * acc / n, 1e3 * (tot - acc) / n))
if os.environ.get("VO_POSE_TRACE") and _lib is not None:
    _lib.vo_debug_pose_trace_dump()
