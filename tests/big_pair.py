"""A reproducible stereo pair of any size from integer arithmetic only (no RNG state, no floating point): the same bytes on
every numpy version.  Used by the 4K regression fixture (tests/golden/make_oracle_4k.py) and the GPU test that replays it."""
import numpy as np


def _hash(ix, iy, seed):
    h = (ix.astype(np.uint64) * np.uint64(374761393) + iy.astype(np.uint64) * np.uint64(668265263)
         + np.uint64(seed) * np.uint64(2246822519)) & np.uint64(0xFFFFFFFF)
    h = ((h ^ (h >> np.uint64(13))) * np.uint64(1274126177)) & np.uint64(0xFFFFFFFF)
    return (h ^ (h >> np.uint64(16))) & np.uint64(0xFFFFFFFF)


def canvas(w, h, seed=7):
    """Blocky value noise at four scales + pixel noise, uint8."""
    y, x = np.mgrid[0:h, 0:w]
    acc = np.zeros((h, w), np.uint64)
    for k, (cell, weight) in enumerate(((64, 96), (16, 64), (4, 48), (1, 24))):
        acc += (_hash(x // cell, y // cell, seed + k) >> np.uint64(24)) * np.uint64(weight)     # 0..255 each
    return np.clip(acc // np.uint64(232) + np.uint64(8), 0, 255).astype(np.uint8)


def pair(w, h, dmin=16, dmax=216, seed=7):
    """Left / right images of a scene whose disparity grows linearly from dmin (top row) to dmax (bottom row): a ground plane.
    A left pixel x matches the right pixel x - d(y)."""
    c = canvas(w + dmax + 8, h, seed)
    d = dmin + (np.arange(h, dtype=np.int64) * (dmax - dmin)) // max(h - 1, 1)
    L = np.ascontiguousarray(c[:, :w])
    R = np.empty_like(L)
    for yy in range(h):
        R[yy] = c[yy, d[yy]:d[yy] + w]
    return L, R, d
