"""BASELINE config 5 (no openVO counterpart): monocular 1920x1080, ORB 8000 keypoints per frame,
8000 x 8000 Hamming kNN, 5000-hypothesis essential-matrix RANSAC.  Prints one JSON line with
op-rates (the stage is not HBM-bound: SURVEY 8(d) asks for Hamming pair-distances/s and residual
evaluations/s)."""
import json, sys, time
sys.path.insert(0, '.')
import numpy as np
from openvo_amd import _native
from openvo_amd.synth import Corridor

c = Corridor("C5")
ctx = _native.Context(0, c.w, c.h, 16, 8000)
frames = [c.pair(k)[0] for k in range(6)]
kps = [ctx.orb_host(f, None, 8000) for f in frames]          # warm-up + data
K4 = [c.f, c.f, c.cx, c.cy]
reps = 5

def timed(fn):
    ctx.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): r = fn()
    ctx.synchronize(); return (time.perf_counter() - t0) / reps, r

t_orb, _ = timed(lambda: ctx.orb_host(frames[1], None, 8000))
a, b = kps[0], kps[1]
t_match, (idx, dist) = timed(lambda: ctx.bf_knn2(a["desc"], b["desc"]))
q, t = ctx.ratio_filter(idx, dist, 0.8)
p1, p2 = a["xy"][q], b["xy"][t]
t_ransac, r = timed(lambda: ctx.ransac_essential(p1, p2, K4, 5000, 1.0, 4321))
nq, nt = len(a["desc"]), len(b["desc"])
# solvePnP hypothesis scoring at the same size: 8000 3-D/2-D correspondences, 5000 hypotheses (40 M residuals)
rng = np.random.default_rng(1)
X = np.stack([rng.uniform(-8, 8, 8000), rng.uniform(-3, 3, 8000), rng.uniform(4, 40, 8000)], 1).astype(np.float32)
Xc = X + np.array([0.05, -0.02, -0.3], np.float32)
uv = np.stack([c.f * Xc[:, 0] / Xc[:, 2] + c.cx, c.f * Xc[:, 1] / Xc[:, 2] + c.cy], 1).astype(np.float32)
uv[:2400] += rng.uniform(-60, 60, (2400, 2)).astype(np.float32)
t_pnp, rp = timed(lambda: ctx.ransac_pnp(X, uv, K4, 5000, 2.0, 4321))
print(json.dumps({"workload": "C5: mono 1920x1080, ORB 8000, kNN 8000x8000, 5000-iter essential RANSAC",
                  "keypoints": [nq, nt], "matches_after_ratio": int(len(q)),
                  "orb_ms": round(1e3 * t_orb, 3), "match_ms": round(1e3 * t_match, 3), "ransac_ms": round(1e3 * t_ransac, 3),
                  "hamming_pair_distances_per_s": round(nq * nt / t_match, 0),
                  "residual_evaluations_per_s": round(5000 * len(q) / t_ransac, 0),
                  "ransac_inliers": r["best_count"],
                  "pnp_ransac_ms": round(1e3 * t_pnp, 3), "pnp_residual_evaluations_per_s": round(5000 * 8000 / t_pnp, 0),
                  "pnp_inliers": rp["best_count"], "note": "host<->device transfers of the seam calls included"}))
