"""Regression fixture at 3840 x 2160, D = 256 (a 3.96 GB cost volume: every 32-bit offset of the volume kernels is past 2^31):
the ORACLE's disparity (oracle/, the CPU restatement -- NOT the reference: OpenCV is absent) of tests/big_pair.pair(3840, 2160),
reduced to a SHA-256 and a few rows (o1_sgbm_4k.npz); and the oracle's ORB keypoints on a 4K image for 2000 and 8000 features,
reduced to digests of every output array (o2_orb_4k.npz: the oracle's selection takes 100 s per call at this size).
Takes a few minutes and ~20 GB of memory on the CPU:  python tests/golden/make_oracle_4k.py [sgbm|orb]
The GPU tests (tests/test_gpu_configs.py::test_sgbm_4k_matches_the_oracle_fixture, ::test_orb_4k_matches_the_oracle_fixture)
replay them."""
import hashlib, os, sys, time
import numpy as np
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import oracle          # noqa: E402
from tests.big_pair import pair    # noqa: E402

W, H = 3840, 2160
PARAMS = dict(minDisparity=0, numDisparities=256, blockSize=5, P1=200, P2=800, disp12MaxDiff=1, preFilterCap=63,
              uniquenessRatio=10, speckleWindowSize=100, speckleRange=2)
ROWS = [0, 7, 540, 1079, 1080, 1621, 2158, 2159]

DT = dict(xy=np.float32, response=np.float32, angle=np.float32, octave=np.int32, desc=np.uint8)


def orb_fixture():
    L, _, _ = pair(W, H, seed=11)
    out = {}
    for n in (2000, 8000):
        t0 = time.time()
        r = oracle.orb_detect_and_compute(L, None, n, cap=4 * n + 4096)
        print("oracle ORB %d: %.1f s, %d keypoints, per level %s" % (n, time.time() - t0, len(r["xy"]), np.bincount(r["octave"], minlength=8)), flush=True)
        out["count_%d" % n] = len(r["xy"])
        out["per_level_%d" % n] = np.bincount(r["octave"], minlength=8)
        for k in ("xy", "response", "angle", "octave", "desc"):
            out["sha256_%s_%d" % (k, n)] = np.frombuffer(hashlib.sha256(np.ascontiguousarray(r[k], dtype=DT[k]).tobytes()).digest(), np.uint8)
        out["head_xy_%d" % n] = r["xy"][:64]
        out["head_desc_%d" % n] = r["desc"][:64]
    np.savez_compressed(os.path.join(HERE, "o2_orb_4k.npz"), w=W, h=H, seed=11, **out)


if __name__ == "__main__":
    what = sys.argv[1] if len(sys.argv) > 1 else "all"
    if what in ("orb", "all"):
        orb_fixture()
    if what == "orb":
        sys.exit(0)
    L, R, d = pair(W, H)
    out = dict(rows=np.array(ROWS, np.int32), w=W, h=H, params=np.array([PARAMS[k] for k in sorted(PARAMS)], np.int32),
               param_names=np.array(sorted(PARAMS)))
    for mode in (0, 1):                                   # MODE_SGBM (5 paths), MODE_HH (8 paths, two passes over the volume)
        t0 = time.time()
        disp = oracle.sgbm_compute(L, R, PARAMS, mode)
        print("oracle mode %d: %.1f s, valid %.3f" % (mode, time.time() - t0, float((disp >= 0).mean())), flush=True)
        out["sha256_mode%d" % mode] = np.frombuffer(hashlib.sha256(disp.tobytes()).digest(), np.uint8)
        out["disp_rows_mode%d" % mode] = disp[ROWS]
        out["valid_fraction_mode%d" % mode] = float((disp >= 0).mean())
        del disp
    np.savez_compressed(os.path.join(HERE, "o1_sgbm_4k.npz"), **out)
