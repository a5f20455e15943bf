import os, sys, numpy as np
sys.path.insert(0, "/root/repo")
from openvo_amd import _native
from openvo_amd.synth import Corridor
for name, mode in (("C1", 0), ("C1", 1), ("C2", 0)):
    c = Corridor(name); L, R = c.pair(4); p = c.sgbm_params(mode)
    out = {}
    for w in ("7", "3"):
        os.environ["VO_DIAG_WAVES"] = w
        ctx = _native.Context(0, c.w, c.h, c.D, 64); ctx.set_sgbm(p, mode)
        out[w] = ctx.sgbm_compute_host(L, R); st = ctx.sgbm_sweep_status(); ctx.close()
        print(name, mode, "waves", w, "status", st)
    print(name, mode, "equal", np.array_equal(out["7"], out["3"]))
