"""host_bound.py for the from-host path: StereoOdometer.run() over numpy pairs (pinned staging + H2D on the engines)."""
import gc, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from openvo_amd import StereoCamera, StereoOdometer
from openvo_amd.synth import Corridor

N = int(os.environ.get("HB_FRAMES", "260"))
c = Corridor("C2")
cam = StereoCamera(c.K(), c.dist(), c.K(), c.dist(), c.rect_params(), c.sgbm_params(), (c.w, c.h), max_keypoints=500)
odo = StereoOdometer(cam, rigidity_threshold=0.1, outlier_threshold=0.02, preprocessed_frames=True)
frames = c.pairs(0, N)
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
for ok in odo.run(frames[:20]): pass
ctx.synchronize()
T.clear()
gc.collect(); gc.disable()
t0 = time.perf_counter()
for ok in odo.run(frames[20:]): pass
ctx.synchronize()
tot = time.perf_counter() - t0
n = N - 20
print("per frame ms: %.3f  (%.1f fps)" % (1e3 * tot / n, n / tot))
acc = 0.0
for k, v in sorted(T.items(), key=lambda kv: -sum(kv[1])):
    acc += sum(v)
    print("%-26s calls/frame=%.2f  ms/frame=%.4f  us/call=%.1f" % (k, len(v) / n, 1e3 * sum(v) / n, 1e6 * sum(v) / len(v)))
print("native calls total ms/frame: %.3f ; python outside native: %.3f" % (1e3 * acc / n, 1e3 * (tot - acc) / n))
