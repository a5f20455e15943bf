"""cProfile of MonoOdometer.update on a staged config-5 stream (host side only; development aid)."""
import cProfile, gc, os, pstats, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from openvo_amd.mono import MonoOdometer
from openvo_amd.synth import Corridor

N = 140
c = Corridor("C5")
Kmat = np.array([[c.f, 0, c.cx], [0, c.f, c.cy], [0, 0, 1.0]])
odo = MonoOdometer(Kmat, (c.w, c.h), nfeatures=8000, ransac_iters=5000, solver=5)
odo.stage_frames([c.pair(k)[0] for k in range(N)])
for i in range(20): odo.update(i)
gc.collect(); gc.disable()
pr = cProfile.Profile()
pr.enable()
for i in range(20, N): odo.update(i)
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(22)
