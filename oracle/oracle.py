"""ctypes binding of the CPU oracle (oracle/libvo_oracle.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg.  Nothing under openvo_amd/ may import this module.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.environ.get("VO_ORACLE_LIB") or os.path.join(_HERE, "libvo_oracle.so")   # VO_ORACLE_LIB: the sanitizer build (make asan)
_SRC = ["src/sgbm.c", "src/imgproc.c", "src/orb.c", "src/match.c", "src/geom.c", "src/ransac.c", "src/fivept.c", "src/pnp.c", "vo_oracle.h"]


def build_oracle(force=False):
    newest = max(os.path.getmtime(os.path.join(_HERE, s)) for s in _SRC)
    if force or not os.path.exists(_LIB) or os.path.getmtime(_LIB) < newest:
        subprocess.check_call(["make", "-C", _HERE, "-B", os.path.basename(_LIB)], stdout=subprocess.DEVNULL)
    return _LIB


class SgbmParams(ctypes.Structure):
    _fields_ = [(n, ctypes.c_int) for n in
                ("minDisparity", "numDisparities", "blockSize", "P1", "P2", "disp12MaxDiff",
                 "preFilterCap", "uniquenessRatio", "speckleWindowSize", "speckleRange", "mode")]


def sgbm_params(d, mode=0):
    return SgbmParams(*[int(d[k]) for k in
                        ("minDisparity", "numDisparities", "blockSize", "P1", "P2", "disp12MaxDiff",
                         "preFilterCap", "uniquenessRatio", "speckleWindowSize", "speckleRange")],
                      int(d.get("mode", mode)))


_lib = None


def lib():
    global _lib
    if _lib is None:
        build_oracle()
        _lib = ctypes.CDLL(_LIB)
        _lib.vo_ref_ratio_filter.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int,
                                             ctypes.c_double, ctypes.c_void_p, ctypes.c_void_p]
        _lib.vo_ref_rigid_clique.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int,
                                             ctypes.c_double, ctypes.c_void_p]
    return _lib


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def _c(a, dt):
    return np.ascontiguousarray(a, dtype=dt)


def bgr2gray(bgr):
    bgr = _c(bgr, np.uint8)
    h, w = bgr.shape[:2]
    out = np.empty((h, w), np.uint8)
    lib().vo_ref_bgr2gray(_p(bgr), w, h, _p(out))
    return out


def remap_bilinear(src, map1, map2):
    src, map1, map2 = _c(src, np.uint8), _c(map1, np.int16), _c(map2, np.uint16)
    h, w = map2.shape
    out = np.empty((h, w), np.uint8)
    lib().vo_ref_remap_bilinear(_p(src), src.shape[1], src.shape[0], _p(map1), _p(map2), w, h, _p(out))
    return out


def sgbm_compute(left, right, params, mode=0, stages=False):
    left, right = _c(left, np.uint8), _c(right, np.uint8)
    h, w = left.shape
    p = sgbm_params(params, mode)
    raw = np.empty((h, w), np.int16)
    med = np.empty((h, w), np.int16)
    fin = np.empty((h, w), np.int16)
    rc = lib().vo_ref_sgbm_compute(_p(left), _p(right), w, h, ctypes.byref(p), _p(raw), _p(med), _p(fin))
    if rc != 0:
        raise ValueError("oracle sgbm: bad parameters")
    return (raw, med, fin) if stages else fin


def sgbm_cost_volume(left, right, params):
    left, right = _c(left, np.uint8), _c(right, np.uint8)
    h, w = left.shape
    p = sgbm_params(params)
    D = p.numDisparities
    minx1 = max(p.minDisparity + D, 0)
    maxx1 = w + min(p.minDisparity, 0)
    C = np.empty((h, maxx1 - minx1, D), np.int16)
    lib().vo_ref_sgbm_cost_volume(_p(left), _p(right), w, h, ctypes.byref(p), _p(C))
    return C


def median3x3_s16(img):
    img = _c(img, np.int16)
    out = np.empty_like(img)
    lib().vo_ref_median3x3_s16(_p(img), img.shape[1], img.shape[0], _p(out))
    return out


def filter_speckles(img, new_val, max_size, max_diff):
    out = _c(img, np.int16).copy()
    lib().vo_ref_filter_speckles(_p(out), out.shape[1], out.shape[0], int(new_val), int(max_size), int(max_diff))
    return out


def orb_detect_and_compute(img, mask, nfeatures=500, blur_mode=0, cap=None):
    """Returns dict(xy, size, angle, response, octave, desc); img/mask may be strided views."""
    img = _c(img, np.uint8)
    h, w = img.shape
    if mask is not None:
        mask = _c(mask, np.uint8)
    cap = cap or (2 * nfeatures + 4096)
    xy = np.empty((cap, 2), np.float32)
    size = np.empty(cap, np.float32)
    angle = np.empty(cap, np.float32)
    resp = np.empty(cap, np.float32)
    octv = np.empty(cap, np.int32)
    desc = np.empty((cap, 32), np.uint8)
    n = ctypes.c_int(0)
    lib().vo_ref_orb_detect_and_compute(_p(img), w, h, w, _p(mask) if mask is not None else None, w,
                                        int(nfeatures), int(blur_mode), _p(xy), _p(size), _p(angle),
                                        _p(resp), _p(octv), _p(desc), cap, ctypes.byref(n))
    n = n.value
    return dict(xy=xy[:n].copy(), size=size[:n].copy(), angle=angle[:n].copy(),
                response=resp[:n].copy(), octave=octv[:n].copy(), desc=desc[:n].copy())


def fast_score_map(img, threshold=20):
    img = _c(img, np.uint8)
    out = np.empty_like(img)
    lib().vo_ref_fast_score_map(_p(img), img.shape[1], img.shape[0], img.shape[1], int(threshold), _p(out))
    return out


def orb_level_size(w, h, level):
    lw, lh = ctypes.c_int(), ctypes.c_int()
    lib().vo_ref_orb_level_size(w, h, level, ctypes.byref(lw), ctypes.byref(lh))
    return lw.value, lh.value


class orb_variant:
    """`with orb_variant(flags):` runs the ORB restatement with the recalled-not-read details of orb.cpp
    switched (bit 0: level size cvRound(cols / scale); bit 1: cosf / sinf) -- only to MEASURE what they weigh."""

    def __init__(self, flags):
        self.flags = int(flags)

    def __enter__(self):
        self.old = lib().vo_ref_orb_get_variant()
        lib().vo_ref_orb_set_variant(self.flags)

    def __exit__(self, *exc):
        lib().vo_ref_orb_set_variant(self.old)


def resize_linear_exact(src, dw, dh):
    src = _c(src, np.uint8)
    out = np.empty((dh, dw), np.uint8)
    lib().vo_ref_resize_linear_exact(_p(src), src.shape[1], src.shape[0], src.shape[1], _p(out), dw, dh, dw)
    return out


def bf_knn2_hamming(q, t):
    q, t = _c(q, np.uint8), _c(t, np.uint8)
    idx = np.empty((len(q), 2), np.int32)
    dist = np.empty((len(q), 2), np.int32)
    lib().vo_ref_bf_knn2_hamming(_p(q), len(q), _p(t), len(t), _p(idx), _p(dist))
    return idx, dist


def ratio_filter(idx, dist, ratio):
    idx, dist = _c(idx, np.int32), _c(dist, np.int32)
    qo = np.empty(len(idx), np.int32)
    to = np.empty(len(idx), np.int32)
    m = lib().vo_ref_ratio_filter(_p(idx), _p(dist), len(idx), float(ratio), _p(qo), _p(to))
    if m < 0:
        raise IndexError("list index out of range")
    return qo[:m].copy(), to[:m].copy()


def reproject_to_3d(disp, Q):
    disp, Q = _c(disp, np.float32), _c(Q, np.float64)
    out = np.empty(disp.shape + (3,), np.float32)
    lib().vo_ref_reproject_to_3d(_p(disp), disp.shape[1], disp.shape[0], _p(Q), _p(out))
    return out


def points3d_at(disp16, Q, roi, xy):
    disp16, Q, xy = _c(disp16, np.int16), _c(Q, np.float64), _c(xy, np.float32).reshape(-1, 2)
    h, w = disp16.shape
    out = np.empty((len(xy), 3), np.float32)
    st = np.empty(len(xy), np.uint8)
    lib().vo_ref_points3d_at(_p(disp16), w, h, _p(Q), int(roi[0]), int(roi[1]), int(roi[2]), int(roi[3]),
                             _p(xy), len(xy), _p(out), _p(st))
    return out, st


def bilinear_at(img3d, xy):
    img3d, xy = _c(img3d, np.float32), _c(xy, np.float32).reshape(-1, 2)
    h, w = img3d.shape[:2]
    out = np.empty((len(xy), 3), np.float32)
    st = np.empty(len(xy), np.uint8)
    lib().vo_ref_bilinear_at(_p(img3d), w, h, _p(xy), len(xy), _p(out), _p(st))
    return out, st


def umeyama(src, dst, force_rotation=True):
    src, dst = _c(src, np.float32), _c(dst, np.float32)
    T = np.empty((3, 4), np.float64)
    s = ctypes.c_double(0)
    rc = lib().vo_ref_umeyama(_p(src), _p(dst), len(src), int(force_rotation), _p(T), ctypes.byref(s))
    if rc == -1:
        raise ValueError("Umeyama algorithm needs at least 3 points for affine transformation estimation.")
    if rc == -2:
        raise ValueError("Points cannot be colinear")
    return T, s.value


def rodrigues(R):
    R = _c(R, np.float64)
    r = np.empty(3, np.float64)
    lib().vo_ref_rodrigues(_p(R), _p(r))
    return r.reshape(3, 1)


def stereo_rectify(K1, D1, K2, D2, img_size, R, T):
    """cv2.stereoRectify(K1, D1, K2, D2, imageSize, R, T) with its Python defaults -> R1, R2, P1, P2, Q, roi1, roi2"""
    K1, K2, R, T = _c(K1, np.float64), _c(K2, np.float64), _c(np.asarray(R, np.float64).reshape(3, 3), np.float64), _c(np.asarray(T, np.float64).reshape(3), np.float64)
    d1 = _c(np.zeros(0) if D1 is None else np.asarray(D1, np.float64).ravel(), np.float64)
    d2 = _c(np.zeros(0) if D2 is None else np.asarray(D2, np.float64).ravel(), np.float64)
    R1, R2, P1, P2, Q = np.empty((3, 3)), np.empty((3, 3)), np.empty((3, 4)), np.empty((3, 4)), np.empty((4, 4))
    roi1, roi2 = np.zeros(4, np.int32), np.zeros(4, np.int32)
    lib().vo_ref_stereo_rectify(_p(K1), _p(d1), len(d1), _p(K2), _p(d2), len(d2), int(img_size[0]), int(img_size[1]), _p(R), _p(T),
                                _p(R1), _p(R2), _p(P1), _p(P2), _p(Q), _p(roi1), _p(roi2))
    return R1, R2, P1, P2, Q, tuple(int(v) for v in roi1), tuple(int(v) for v in roi2)


def init_undistort_rectify_map(K, dist, R, P, img_size):
    """cv2.initUndistortRectifyMap(K, dist, R, P, size, CV_16SC2) -> (map1 int16 HxWx2, map2 uint16 HxW)"""
    K, R, P = _c(K, np.float64), _c(np.eye(3) if R is None else R, np.float64), _c(P, np.float64)
    d = _c(np.zeros(0) if dist is None else np.asarray(dist, np.float64).ravel(), np.float64)
    w, h = int(img_size[0]), int(img_size[1])
    m1, m2 = np.empty((h, w, 2), np.int16), np.empty((h, w), np.uint16)
    Pm = np.zeros((3, 4))
    Pm[:, :P.shape[1]] = P[:3]
    lib().vo_ref_init_undistort_rectify_map(_p(K), _p(d), len(d), _p(R), _p(_c(Pm, np.float64)), w, h, _p(m1), _p(m2))
    return m1, m2


def rigid_clique(prev, cur, thr):
    prev, cur = _c(prev, np.float32), _c(cur, np.float32)
    mask = np.zeros(len(cur), np.int64)
    lib().vo_ref_rigid_clique(_p(prev), _p(cur), len(cur), float(thr), _p(mask))
    return mask


def svd3(A):
    A = _c(A, np.float64)
    U, w, Vt = np.empty((3, 3)), np.empty(3), np.empty((3, 3))
    lib().vo_ref_svd3(_p(A), _p(U), _p(w), _p(Vt))
    return U, w, Vt


def poly10_roots_unit(c):
    """real roots inside [-1, 1] of sum c[k] z^k (11 coefficients), increasing -- the five-point solver's root finder"""
    c = _c(c, np.float64).reshape(11)
    out = np.zeros(10, np.float64)
    f = lib().vo_ref_poly10_roots_unit
    f.argtypes = [ctypes.c_void_p] * 2
    f.restype = ctypes.c_int
    return out[:f(_p(c), _p(out))].copy()


def essential_5pt(x1, x2):
    """5 normalised correspondences -> list of candidate essential matrices (<= 10, unit Frobenius norm)"""
    x1, x2 = _c(x1, np.float64).reshape(5, 2), _c(x2, np.float64).reshape(5, 2)
    out = np.zeros(90, np.float64)
    f = lib().vo_ref_essential_5pt
    f.argtypes = [ctypes.c_void_p] * 3
    f.restype = ctypes.c_int
    n = f(_p(x1), _p(x2), _p(out))
    return [out[9 * k:9 * k + 9].reshape(3, 3).copy() for k in range(n)]


def ransac_essential(p1, p2, K4, iters=5000, thr=1.0, seed=4321, solver=8):
    """-> dict(E 3x3, mask uint8 n, counts int32 iters, best_iter, best_count); solver 8 (eight-point) or 5 (five-point
    + a sixth correspondence to pick among its solutions)"""
    p1, p2, K4 = _c(p1, np.float32).reshape(-1, 2), _c(p2, np.float32).reshape(-1, 2), _c(K4, np.float64)
    n = len(p1)
    E = np.zeros(9, np.float64)
    mask = np.zeros(n, np.uint8)
    counts = np.zeros(iters, np.int32)
    bi = ctypes.c_int(-1)
    f = lib().vo_ref_ransac_essential5 if solver == 5 else lib().vo_ref_ransac_essential
    f.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_float,
                  ctypes.c_uint32, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
    best = f(_p(p1), _p(p2), n, _p(K4), int(iters), float(thr), int(seed), _p(E), _p(mask), _p(counts), ctypes.byref(bi))
    if best < 0:
        raise ValueError("ransac_essential needs at least %d correspondences" % (6 if solver == 5 else 8))
    return dict(E=E.reshape(3, 3), mask=mask, counts=counts, best_iter=bi.value, best_count=best)


def p3p(bearings, points):
    """3 unit bearings + 3 points -> list of (R 3x3, t 3) candidate poses (<= 4)"""
    y, x = _c(bearings, np.float64).reshape(3, 3), _c(points, np.float64).reshape(3, 3)
    R, t = np.zeros(36), np.zeros(12)
    f = lib().vo_ref_p3p
    f.argtypes = [ctypes.c_void_p] * 4
    f.restype = ctypes.c_int
    n = f(_p(y), _p(x), _p(R), _p(t))
    return [(R[9 * k:9 * k + 9].reshape(3, 3).copy(), t[3 * k:3 * k + 3].copy()) for k in range(n)]


def ransac_pnp(pts3d, pts2d, K4, iters=5000, thr=2.0, seed=4321):
    """-> dict(Rt 3x4, mask uint8 n, counts int32 iters, best_iter, best_count)"""
    X, uv, K4 = _c(pts3d, np.float32).reshape(-1, 3), _c(pts2d, np.float32).reshape(-1, 2), _c(K4, np.float64)
    n = len(X)
    if len(uv) != n:
        raise ValueError("pts3d / pts2d lengths differ")
    Rt = np.zeros(12, np.float64)
    mask = np.zeros(n, np.uint8)
    counts = np.zeros(iters, np.int32)
    bi = ctypes.c_int(-1)
    f = lib().vo_ref_ransac_pnp
    f.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_float,
                  ctypes.c_uint32, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
    f.restype = ctypes.c_int
    best = f(_p(X), _p(uv), n, _p(K4), int(iters), float(thr), int(seed), _p(Rt), _p(mask), _p(counts), ctypes.byref(bi))
    if best < 0:
        raise ValueError("ransac_pnp needs at least 4 correspondences")
    return dict(Rt=Rt.reshape(3, 4), mask=mask, counts=counts, best_iter=bi.value, best_count=best)
