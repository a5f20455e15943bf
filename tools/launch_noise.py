"""Do kernel boundaries of OTHER streams cost the SGBM engines throughput?  (development aid)
Disparity-only loop of tools/stage_ablation.py, once alone and once beside a thread that launches tiny kernels (one workgroup,
a few instructions) on a stream of its own as fast as it can.  Every kernel end is a release point (L2 write-back)."""
import os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from openvo_amd import _native
from openvo_amd.synth import Corridor

c = Corridor("C2")
N = 480
ctx = _native.Context(0, c.w, c.h, c.D, 500)
ctx.set_sgbm(c.sgbm_params(0), 0)
ctx.stage_pairs([c.pair(i) for i in range(24)])

def loop():
    for i in range(24):
        ctx.prefetch_staged_pair(i % 24, i % 24, True)
    ctx.synchronize()
    t0 = time.perf_counter()
    for i in range(N):
        ctx.prefetch_staged_pair(i % 24, i % 24, True)
    ctx.synchronize()
    return N / (time.perf_counter() - t0)

print("alone: %.0f pairs/s" % loop())
stop, count = False, [0]
def noise(sleep):
    s = torch.cuda.Stream()
    x = torch.zeros(64, device="cuda")
    with torch.cuda.stream(s):
        while not stop:
            x.add_(1.0)
            count[0] += 1
            if sleep:
                time.sleep(sleep)
for sleep in (0.0, 0.00005, 0.0002):
    stop, count[0] = False, 0
    th = threading.Thread(target=noise, args=(sleep,))
    th.start()
    time.sleep(0.2)
    c0, t0 = count[0], time.perf_counter()
    r = loop()
    rate = (count[0] - c0) / (time.perf_counter() - t0)
    stop = True
    th.join()
    torch.cuda.synchronize()
    print("beside %.0f tiny launches/s: %.0f pairs/s" % (rate, r))
