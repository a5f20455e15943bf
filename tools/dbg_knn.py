import sys, os, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import numpy as np
from openvo_amd import _native
from oracle import oracle as O
ctx = _native.Context(0, 640, 480, 64, 8000)
rng = np.random.default_rng(1)
for nq, nt in ((64, 16), (64, 64), (500, 500), (512, 512), (1000, 2000), (8000, 8000), (8012, 8030), (3, 1), (5, 2), (700, 17)):
    q = rng.integers(0, 256, (nq, 32), dtype=np.uint8); t = rng.integers(0, 256, (nt, 32), dtype=np.uint8)
    gi, gd = ctx.bf_knn2(q, t); ri, rd = O.bf_knn2_hamming(q, t)
    bad = np.nonzero((gi != ri).any(axis=1) | (gd != rd).any(axis=1))[0]
    print(nq, nt, "bad rows", len(bad), bad[:10], flush=True)
    for b in bad[:3]:
        print("   row", b, "got", gi[b], gd[b], "want", ri[b], rd[b])
