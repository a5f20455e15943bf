"""Where does the host thread spend a monocular (config 5) frame?  Same idea as host_bound.py: wall timers around the native
calls of MonoOdometer.update -- enqueue-only calls vs calls that wait for the GPU.  VO_MONO_SPECULATE / VO_MONO_LOOKAHEAD apply."""
import gc, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from openvo_amd.mono import MonoOdometer
from openvo_amd.synth import Corridor

N = int(os.environ.get("HB_FRAMES", "160"))
c = Corridor("C5")
Kmat = np.array([[c.f, 0, c.cx], [0, c.f, c.cy], [0, 0, 1.0]])
odo = MonoOdometer(Kmat, (c.w, c.h), nfeatures=8000, ransac_iters=5000, solver=5)
odo.stage_frames([c.pair(k)[0] for k in range(N)])
ctx = odo._ctx
T = {}
def wrap(obj, name):
    f = getattr(obj, name)
    def g(*a, **k):
        t0 = time.perf_counter(); r = f(*a, **k); T.setdefault(name, []).append(time.perf_counter() - t0); return r
    setattr(obj, name, g)
for n in dir(ctx):
    if not n.startswith("_") and callable(getattr(ctx, n)) and n not in ("close",):
        wrap(ctx, n)
for i in range(20): odo.update(i)
T.clear()
gc.collect(); gc.disable()
t0 = time.perf_counter()
for i in range(20, N): odo.update(i)
ctx.synchronize()
tot = time.perf_counter() - t0
n = N - 20
print("per frame ms: %.3f  (%.1f fps)  speculation %s" % (1e3 * tot / n, n / tot, odo.speculation))
acc = 0.0
for k, v in sorted(T.items(), key=lambda kv: -sum(kv[1])):
    acc += sum(v)
    print("%-26s calls/frame=%.2f  ms/frame=%.4f  us/call=%.1f" % (k, len(v) / n, 1e3 * sum(v) / n, 1e6 * sum(v) / len(v)))
print("native calls total ms/frame: %.3f ; python outside native: %.3f" % (1e3 * acc / n, 1e3 * (tot - acc) / n))
