"""Debug aid: raster SGBM vs the oracle on one small pair; prints where they differ and the sweep status."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from openvo_amd import _native
from openvo_amd.synth import Corridor
from oracle import oracle as O
O.build_oracle()
name = sys.argv[1] if len(sys.argv) > 1 else "T0"
mode = int(sys.argv[2]) if len(sys.argv) > 2 else 0
c = Corridor(name)
L, R = c.pair(3)
p = c.sgbm_params()
p["speckleWindowSize"] = 0
ctx = _native.Context(0, max(c.w, 704), max(c.h, 512), max(64, c.D), 1000)
ctx.set_sgbm(p, mode)
got = ctx.sgbm_compute_host(L, R)
print("raster status", ctx.sgbm_raster_status())
ref = O.sgbm_compute(L, R, p, mode)
bad = got != ref
print(name, "shape", got.shape, "D", c.D, "bad", int(bad.sum()), "of", bad.size)
if bad.any():
    ys, xs = np.nonzero(bad)
    print("rows with errors: first %d last %d count %d" % (ys.min(), ys.max(), len(np.unique(ys))))
    print("cols with errors: first %d last %d" % (xs.min(), xs.max()))
    per_row = bad.sum(1)
    print("bad per row (first 40):", per_row[:40].tolist())
    y = ys.min()
    print("row", y, "got", got[y, c.D:c.D + 24].tolist())
    print("row", y, "ref", ref[y, c.D:c.D + 24].tolist())
