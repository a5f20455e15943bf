"""RANSAC essential-matrix (BASELINE config 5) and solvePnP estimation on the GPU.

Not part of openVO -- the reference fits its pose in closed form.  Provided as an
extra, with its own CPU restatement for parity; `StereoOdometer` never calls it.
"""
import numpy as np

from . import _native


def find_essential_mat(points1, points2, K, iters=5000, threshold=1.0, seed=4321, context=None, solver=8):
    """points: (n, 2) pixel coordinates of matched keypoints in two views; K: 3x3 intrinsics.
    Returns (E 3x3 float64, inlier mask uint8 (n,), info dict).  Hypotheses come from 8-point minimal
    sets (solver=8) or from the five-point minimal solver with a sixth correspondence choosing among
    its solutions (solver=5) -- hash RNG, reproducible for a given seed; the winner has the most
    Sampson inliers (threshold in pixels), ties going to the earliest hypothesis."""
    K = np.asarray(K, np.float64)
    own = context is None
    ctx = context or _native.Context(0, 64, 64, 16, 64)
    try:
        r = ctx.ransac_essential(points1, points2, [K[0, 0], K[1, 1], K[0, 2], K[1, 2]], iters, threshold, seed, solver=solver)
    finally:
        if own:
            ctx.close()
    return r["E"], r["mask"], dict(best_iter=r["best_iter"], best_count=r["best_count"])


def solve_pnp_ransac(points3d, points2d, K, iters=5000, threshold=2.0, seed=4321, context=None):
    """points3d: (n, 3) points in the first view's frame; points2d: (n, 2) pixel positions of the same
    features in the second view; K: 3x3 intrinsics.  Returns (R 3x3, t 3, inlier mask uint8 (n,), info).
    Hypotheses: P3P on hash-sampled minimal sets, a fourth sample disambiguating; the winner has the most
    points reprojecting within `threshold` pixels, ties going to the earliest hypothesis.  x_cam = R X + t."""
    K = np.asarray(K, np.float64)
    own = context is None
    ctx = context or _native.Context(0, 64, 64, 16, 64)
    try:
        r = ctx.ransac_pnp(points3d, points2d, [K[0, 0], K[1, 1], K[0, 2], K[1, 2]], iters, threshold, seed)
    finally:
        if own:
            ctx.close()
    return r["Rt"][:, :3].copy(), r["Rt"][:, 3].copy(), r["mask"], dict(best_iter=r["best_iter"], best_count=r["best_count"])
