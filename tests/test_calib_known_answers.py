"""StereoCamera.__init__ without cv2 (openvo_amd/calib.py restating cv2.stereoRectify and
cv2.initUndistortRectifyMap(CV_16SC2), reference stereo_camera.py:17-22).

cv2 is not installed here, so these are known answers derived by hand from OpenCV 4.x's formulas
(calib3d/src/calibration.cpp cvStereoRectify / icvGetRectangles, imgproc undistort) for rigs where they
close: closed forms for the undistorted rigs, a scalar step-by-step evaluation for the rotated + distorted
one, and the geometric property rectification exists for (matching rows, maps invert the lens model).
tests/test_cv2_crosscheck.py compares the same rigs with a real cv2 wherever one is importable."""
import math

import numpy as np

from openvo_amd import calib

W, H = 640, 480


def _K(fx, fy, cx, cy):
    return np.array([[fx, 0, cx], [0, fy, cy], [0, 0, 1.0]])


def test_unequal_focal_lengths_closed_form():
    """R = I, T along -x, no distortion: fc_new = (fy1 + fy2) / 2 (OpenCV >= 3.4.8), new principal point =
    mean over both cameras of (n-1)/2 - fc_new ((n-1)/2 - c_k) / f_k, Q and P2 from them, ROI from the
    rectified image borders, maps affine."""
    K1, K2 = _K(500.0, 510.0, 322.0, 241.0), _K(520.0, 530.0, 318.0, 238.0)
    B = 0.25
    R1, R2, P1, P2, Q, roi1, roi2 = calib.stereo_rectify(K1, None, K2, None, (W, H), np.eye(3), [-B, 0, 0])
    assert np.allclose(R1, np.eye(3), atol=1e-15) and np.allclose(R2, np.eye(3), atol=1e-15)
    fc = (510.0 + 530.0) / 2
    assert P1[0, 0] == fc and P1[1, 1] == fc and P2[0, 0] == fc and P2[1, 1] == fc

    def centre(n, c1, f1, c2, f2):
        return 0.5 * (((n - 1) / 2 - fc * ((n - 1) / 2 - c1) / f1) + ((n - 1) / 2 - fc * ((n - 1) / 2 - c2) / f2))
    cx, cy = centre(W, 322.0, 500.0, 318.0, 520.0), centre(H, 241.0, 510.0, 238.0, 530.0)
    # the four corners travel through float32 (CvPoint2D32f): a few 1e-5 px of rounding
    assert abs(P1[0, 2] - cx) < 2e-4 and abs(P1[1, 2] - cy) < 2e-4
    assert P2[0, 2] == P1[0, 2] and P2[1, 2] == P1[1, 2]            # CALIB_ZERO_DISPARITY
    assert abs(P2[0, 3] + B * fc) < 1e-12 and P2[1, 3] == 0
    assert np.allclose(Q, [[1, 0, 0, -P1[0, 2]], [0, 1, 0, -P1[1, 2]], [0, 0, 0, fc], [0, 0, 1 / B, 0]], atol=1e-12)
    # valid ROI: rectified positions of the source image border, inner rectangle, ceil / floor
    for K, roi in ((K1, roi1), (K2, roi2)):
        x0 = fc * (0 - K[0, 2]) / K[0, 0] + P1[0, 2]
        x1 = fc * (W - 1 - K[0, 2]) / K[0, 0] + P1[0, 2]
        y0 = fc * (0 - K[1, 2]) / K[1, 1] + P1[1, 2]
        y1 = fc * (H - 1 - K[1, 2]) / K[1, 1] + P1[1, 2]
        ex, ey = math.ceil(x0), math.ceil(y0)
        ew, eh = math.floor(np.float32(np.float32(x1) - np.float32(x0))), math.floor(np.float32(np.float32(y1) - np.float32(y0)))
        want = (max(ex, 0), max(ey, 0), min(ex + ew, W) - max(ex, 0), min(ey + eh, H) - max(ey, 0))
        assert roi == want, (roi, want)
    # maps: source = f_k (dst - c_new) / fc + c_k, stored as Q5 fixed point
    m1, m2 = calib.init_undistort_rectify_map(K2, None, R2, P2, (W, H))
    for (i, j) in [(0, 0), (0, W - 1), (H - 1, 0), (H - 1, W - 1), (237, 318), (100, 555), (479, 1), (1, 638)]:
        u = 520.0 * (j - P2[0, 2]) / fc + 318.0
        v = 530.0 * (i - P2[1, 2]) / fc + 238.0
        iu, iv = int(np.rint(u * 32)), int(np.rint(v * 32))
        assert abs(u * 32 - np.floor(u * 32) - 0.5) > 1e-6 and abs(v * 32 - np.floor(v * 32) - 0.5) > 1e-6   # no rounding tie
        assert (m1[i, j, 0], m1[i, j, 1]) == (iu >> 5, iv >> 5), (i, j)
        assert m2[i, j] == (iv & 31) * 32 + (iu & 31), (i, j)


def test_negative_k1_does_not_shrink_the_focal_length():
    """The pre-3.4.8 rule (fc *= 1 + k1 (nx^2 + ny^2) / (4 fc^2) for k1 < 0) is gone in every OpenCV that has
    estimateAffine3D(force_rotation) -- the reference's floor (>= 4.5.5)."""
    K = _K(400.0, 400.0, 319.5, 239.5)
    d = [-0.25, 0.05, 0.0, 0.0, 0.0]
    _, _, P1, P2, Q, roi1, _ = calib.stereo_rectify(K, d, K, d, (W, H), np.eye(3), [-0.12, 0, 0])
    assert P1[0, 0] == 400.0 and P2[1, 1] == 400.0 and Q[2, 3] == 400.0
    assert abs(P2[0, 3] + 0.12 * 400.0) < 1e-12
    # barrel distortion stretches the rectified image past the frame: the valid region is the whole frame
    assert roi1 == (0, 0, W, H)
    # pincushion distortion (k1 > 0) pulls the rectified border inside: a strictly interior, centred rectangle
    dp = [0.2, 0.0, 0.0, 0.0, 0.0]
    _, _, Pp, _, _, roip, roip2 = calib.stereo_rectify(K, dp, K, dp, (W, H), np.eye(3), [-0.12, 0, 0])
    x, y, w, h = roip
    assert Pp[0, 0] == 400.0 and roip == roip2
    assert x > 0 and y > 0 and x + w < W and y + h < H
    assert abs((x + x + w) / 2 - 319.5) <= 1.0 and abs((y + y + h) / 2 - 239.5) <= 1.0


# ---- scalar evaluation of cvStereoRectify for a rotated + distorted rig --------------------------------------
def _rodrigues(om):
    th = math.sqrt(sum(v * v for v in om))
    if th < 1e-300:
        return [[1.0, 0, 0], [0, 1.0, 0], [0, 0, 1.0]]
    k = [v / th for v in om]
    c, s = math.cos(th), math.sin(th)
    Kx = [[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]]
    return [[c * (i == j) + (1 - c) * k[i] * k[j] + s * Kx[i][j] for j in range(3)] for i in range(3)]


def _mm(A, B):
    return [[sum(A[i][k] * B[k][j] for k in range(3)) for j in range(3)] for i in range(3)]


def _mv(A, v):
    return [sum(A[i][k] * v[k] for k in range(3)) for i in range(3)]


def _tr(A):
    return [[A[j][i] for j in range(3)] for i in range(3)]


def _distort(x, y, k):
    r2 = x * x + y * y
    kr = (1 + ((k[4] * r2 + k[1]) * r2 + k[0]) * r2)
    return (x * kr + 2 * k[2] * x * y + k[3] * (r2 + 2 * x * x), y * kr + k[2] * (r2 + 2 * y * y) + 2 * k[3] * x * y)


def _undistort(u, v, K, k):
    x0, y0 = (u - K[0][2]) / K[0][0], (v - K[1][2]) / K[1][1]
    x, y = x0, y0
    for _ in range(5):
        r2 = x * x + y * y
        ic = 1.0 / (1 + ((k[4] * r2 + k[1]) * r2 + k[0]) * r2)
        dx = 2 * k[2] * x * y + k[3] * (r2 + 2 * x * x)
        dy = k[2] * (r2 + 2 * y * y) + 2 * k[3] * x * y
        x, y = (x0 - dx) * ic, (y0 - dy) * ic
    return x, y


RIG = dict(K1=[[498.0, 0, 325.0], [0, 502.0, 236.0], [0, 0, 1]], K2=[[507.0, 0, 314.0], [0, 511.0, 244.0], [0, 0, 1]],
           d1=[-0.11, 0.03, 0.0007, -0.0004, 0.002], d2=[-0.09, 0.015, -0.0003, 0.0006, -0.001],
           om=[0.012, -0.021, 0.008], T=[-0.2, 0.004, -0.003])


def test_rotated_distorted_rig_against_scalar_evaluation():
    K1, K2, d1, d2, T = RIG["K1"], RIG["K2"], RIG["d1"], RIG["d2"], RIG["T"]
    Rm = _rodrigues(RIG["om"])
    R1, R2, P1, P2, Q, roi1, roi2 = calib.stereo_rectify(np.array(K1), d1, np.array(K2), d2, (W, H), np.array(Rm), T)
    # Bouguet: both cameras turn half way, then a common rotation brings the baseline onto the x axis
    r_r = _rodrigues([-0.5 * v for v in RIG["om"]])
    t = _mv(r_r, T)
    assert abs(t[0]) > abs(t[1])
    nt = math.sqrt(sum(v * v for v in t))
    uu = [-1.0, 0.0, 0.0]                       # t[0] < 0
    ww = [t[1] * uu[2] - t[2] * uu[1], t[2] * uu[0] - t[0] * uu[2], t[0] * uu[1] - t[1] * uu[0]]
    nw = math.sqrt(sum(v * v for v in ww))
    ww = [v * math.acos(abs(t[0]) / nt) / nw for v in ww]
    wR = _rodrigues(ww)
    eR1, eR2 = _mm(wR, _tr(r_r)), _mm(wR, r_r)
    assert np.allclose(R1, eR1, atol=1e-12) and np.allclose(R2, eR2, atol=1e-12)
    tt = _mv(eR2, T)
    assert abs(tt[1]) < 1e-12 and abs(tt[2]) < 1e-12 and tt[0] < 0
    fc = (502.0 + 511.0) / 2
    cc = []
    for K, d, Rk in ((K1, d1, eR1), (K2, d2, eR2)):
        sx = sy = 0.0
        for (u, v) in ((0, 0), (W - 1, 0), (0, H - 1), (W - 1, H - 1)):
            x, y = _undistort(u, v, K, d)
            X = _mv(Rk, [x, y, 1.0])
            sx += fc * X[0] / X[2]
            sy += fc * X[1] / X[2]
        cc.append(((W - 1) / 2 - sx / 4, (H - 1) / 2 - sy / 4))
    cx, cy = (cc[0][0] + cc[1][0]) / 2, (cc[0][1] + cc[1][1]) / 2
    assert P1[0, 0] == fc and P2[1, 1] == fc
    assert abs(P1[0, 2] - cx) < 2e-4 and abs(P1[1, 2] - cy) < 2e-4 and P2[0, 2] == P1[0, 2] and P2[1, 2] == P1[1, 2]
    assert abs(P2[0, 3] - tt[0] * fc) < 1e-9
    assert np.allclose(Q, [[1, 0, 0, -P1[0, 2]], [0, 1, 0, -P1[1, 2]], [0, 0, 0, fc], [0, 0, -1 / tt[0], 0]], atol=1e-12)
    # ROI: inscribed rectangle of the 9x9 grid of source points pushed through undistort -> R -> P
    for K, d, Rk, roi in ((K1, d1, eR1, roi1), (K2, d2, eR2, roi2)):
        g = {}
        for a in range(9):
            for b in range(9):
                x, y = _undistort(float(np.float32(b * (W - 1) / 8)), float(np.float32(a * (H - 1) / 8)), K, d)
                X = _mv(Rk, [x, y, 1.0])
                g[a, b] = (np.float32(fc * X[0] / X[2] + P1[0, 2]), np.float32(fc * X[1] / X[2] + P1[1, 2]))
        ix0, ix1 = max(g[a, 0][0] for a in range(9)), min(g[a, 8][0] for a in range(9))
        iy0, iy1 = max(g[0, b][1] for b in range(9)), min(g[8, b][1] for b in range(9))
        ex, ey, ew, eh = math.ceil(ix0), math.ceil(iy0), math.floor(np.float32(ix1 - ix0)), math.floor(np.float32(iy1 - iy0))
        want = (max(ex, 0), max(ey, 0), min(ex + ew, W) - max(ex, 0), min(ey + eh, H) - max(ey, 0))
        assert roi == want, (roi, want)
        assert 0 < roi[2] <= W and 0 < roi[3] <= H
    # maps at a dozen pixels incl. the borders: dst pixel -> normalised rectified ray -> R^T -> lens model -> K
    for K, d, Rk, P in ((K1, d1, eR1, P1), (K2, d2, eR2, P2)):
        m1, m2 = calib.init_undistort_rectify_map(np.array(K), d, np.array(Rk), P, (W, H))
        assert m1.shape == (H, W, 2) and m1.dtype == np.int16 and m2.shape == (H, W) and m2.dtype == np.uint16
        for (i, j) in [(0, 0), (0, W - 1), (H - 1, 0), (H - 1, W - 1), (0, 320), (240, 0), (240, W - 1), (H - 1, 320),
                       (240, 320), (17, 601), (333, 45), (455, 500)]:
            ray = _mv(_tr(Rk), [(j - P[0, 2]) / fc, (i - P[1, 2]) / fc, 1.0])
            xd, yd = _distort(ray[0] / ray[2], ray[1] / ray[2], d)
            u, v = K[0][0] * xd + K[0][2], K[1][1] * yd + K[1][2]
            iu, iv = int(np.rint(u * 32)), int(np.rint(v * 32))
            if min(abs(u * 32 - math.floor(u * 32) - 0.5), abs(v * 32 - math.floor(v * 32) - 0.5)) < 1e-6:
                continue                         # a rounding tie would test float noise, not the formula
            assert (int(m1[i, j, 0]), int(m1[i, j, 1])) == (iu >> 5, iv >> 5), (i, j)
            assert int(m2[i, j]) == (iv & 31) * 32 + (iu & 31), (i, j)


def test_rectification_aligns_rows_and_maps_invert_the_lens():
    """What rectification is for: a 3-D point lands on the same row of both rectified images, at a column
    difference of -P2[0,3] / Z; and the remap table of each camera points from that rectified pixel back to
    the pixel where the (distorted) camera really sees the point."""
    K1, K2, d1, d2 = np.array(RIG["K1"]), np.array(RIG["K2"]), RIG["d1"], RIG["d2"]
    Rm, T = np.array(_rodrigues(RIG["om"])), np.array(RIG["T"])
    R1, R2, P1, P2, Q, _, _ = calib.stereo_rectify(K1, d1, K2, d2, (W, H), Rm, T)
    maps = [calib.init_undistort_rectify_map(K1, d1, R1, P1, (W, H)), calib.init_undistort_rectify_map(K2, d2, R2, P2, (W, H))]
    rng = np.random.default_rng(3)
    n_checked = 0
    for _ in range(4000):
        X1 = np.array([rng.uniform(-1.5, 1.5), rng.uniform(-1.0, 1.0), rng.uniform(2.0, 12.0)])
        X2 = Rm @ X1 + T
        rect = []
        for X, Rk, P in ((X1, R1, P1), (X2, R2, P2)):
            Xr = Rk @ X
            rect.append(np.array([P[0, 0] * Xr[0] / Xr[2] + P[0, 2], P[1, 1] * Xr[1] / Xr[2] + P[1, 2], Xr[2]]))   # X is in this camera's own frame
        assert abs(rect[0][1] - rect[1][1]) < 1e-9                                  # same row
        assert abs((rect[0][0] - rect[1][0]) + P2[0, 3] / rect[0][2]) < 1e-9          # disparity = f B / Z
        # reprojectImageTo3D's formula with Q gives the point back (in the rectified left frame)
        d = rect[0][0] - rect[1][0]
        Wq = Q[3, 2] * d + Q[3, 3]
        back = np.array([rect[0][0] + Q[0, 3], rect[0][1] + Q[1, 3], Q[2, 3]]) / Wq
        assert np.allclose(back, R1 @ X1, atol=1e-9)
        for X, K, dd, (m1, m2), r in ((X1, K1, d1, maps[0], rect[0]), (X2, K2, d2, maps[1], rect[1])):
            j, i = int(round(r[0])), int(round(r[1]))
            if not (1 <= j < W - 1 and 1 <= i < H - 1 and abs(r[0] - j) < 0.02 and abs(r[1] - i) < 0.02):
                continue
            xd, yd = _distort(X[0] / X[2], X[1] / X[2], dd)
            u, v = K[0, 0] * xd + K[0, 2], K[1, 1] * yd + K[1, 2]
            mu = m1[i, j, 0] + (int(m2[i, j]) & 31) / 32.0
            mv = m1[i, j, 1] + (int(m2[i, j]) >> 5) / 32.0
            assert abs(mu - u) < 0.05 and abs(mv - v) < 0.05
            n_checked += 1
    assert n_checked >= 3


def test_vertical_rig_uses_the_other_axis():
    """|T_y| > |T_x|: idx = 1, fc_new from K[0,0], P2[1,3] carries the baseline."""
    K = _K(450.0, 470.0, 320.0, 240.0)
    R1, R2, P1, P2, Q, _, _ = calib.stereo_rectify(K, None, K, None, (W, H), np.eye(3), [0.0, -0.15, 0.0])
    assert P1[0, 0] == 450.0 and P2[0, 3] == 0 and abs(P2[1, 3] + 0.15 * 450.0) < 1e-12
    assert abs(Q[3, 2] - 1 / 0.15) < 1e-12


# ---- the product's calib.py against its checker: the C restatement in oracle/src/calib.c ----------------------------------
def _rigs():
    rot = _rodrigues(RIG["om"])
    yield "rotated_distorted", np.array(RIG["K1"]), RIG["d1"], np.array(RIG["K2"]), RIG["d2"], np.array(rot), RIG["T"], (W, H)
    yield "ideal", _K(718.856, 718.856, 640.0, 360.0), np.zeros(5), _K(718.856, 718.856, 640.0, 360.0), np.zeros(5), np.eye(3), [-0.537, 0, 0], (1280, 720)
    yield "vertical", _K(500.0, 505.0, 322.0, 241.0), [0.05, -0.02, 0, 0, 0], _K(515.0, 512.0, 318.0, 238.0), [0.03, 0.01, 0.001, -0.001, 0],\
        np.array(_rodrigues([0.004, 0.006, -0.011])), [0.003, -0.18, 0.002], (W, H)
    yield "eight_coefficients", _K(610.0, 612.0, 300.0, 250.0), [-0.2, 0.05, 0.001, 0.0005, -0.004, 0.01, -0.002, 0.0003], \
        _K(605.0, 609.0, 330.0, 236.0), [-0.18, 0.04, -0.0007, 0.0002, -0.003, 0.008, -0.001, 0.0002], \
        np.array(_rodrigues([-0.02, 0.03, 0.01])), [0.25, -0.006, 0.004], (W, H)
    yield "no_distortion_given", _K(400.0, 400.0, 160.0, 120.0), None, _K(400.0, 400.0, 160.0, 120.0), None, np.eye(3), [-0.12, 0, 0], (320, 240)


def test_calib_py_equals_the_c_oracle(oracle):
    """openvo_amd/calib.py is the product; oracle/src/calib.c restates cvStereoRectify / icvGetRectangles /
    initUndistortRectifyMap + convertMaps independently (scalar C, pixel by pixel).  R1, R2, P1, P2, Q agree to 1e-12,
    the valid ROIs exactly, the fixed-point maps everywhere except isolated rounding ties (<= 1 LSB, < 1e-4 of the pixels).
    Reference call sites: stereo_camera.py:17-22."""
    n = 0
    for name, K1, d1, K2, d2, R, T, size in _rigs():
        got = calib.stereo_rectify(K1, d1, K2, d2, size, R, T)
        ref = oracle.stereo_rectify(K1, d1, K2, d2, size, R, T)
        for g, r, what in zip(got[:5], ref[:5], ("R1", "R2", "P1", "P2", "Q")):
            assert np.allclose(g, r, rtol=0, atol=1e-12), (name, what, np.abs(np.asarray(g) - r).max())
        assert tuple(got[5]) == tuple(ref[5]) and tuple(got[6]) == tuple(ref[6]), (name, got[5:], ref[5:])
        assert got[5][2] > 0 and got[5][3] > 0
        for K, d, Rk, P in ((K1, d1, got[0], got[2]), (K2, d2, got[1], got[3])):
            m1, m2 = calib.init_undistort_rectify_map(K, d, Rk, P, size)
            o1, o2 = oracle.init_undistort_rectify_map(K, d, Rk, P, size)
            assert m1.shape == o1.shape and m2.shape == o2.shape and m1.dtype == o1.dtype and m2.dtype == o2.dtype
            # compare as Q5 fixed-point positions: (map1 << 5) + fraction
            fu, fv = m1[..., 0].astype(np.int64) * 32 + (m2 & 31), m1[..., 1].astype(np.int64) * 32 + (m2 >> 5)
            gu, gv = o1[..., 0].astype(np.int64) * 32 + (o2 & 31), o1[..., 1].astype(np.int64) * 32 + (o2 >> 5)
            du, dv = np.abs(fu - gu), np.abs(fv - gv)
            assert du.max() <= 1 and dv.max() <= 1, (name, int(du.max()), int(dv.max()))
            assert ((du + dv) > 0).mean() < 1e-4, (name, float(((du + dv) > 0).mean()))
            n += 1
    assert n == 10


def test_calib_fuzz_py_against_the_c_oracle(oracle):
    """16 random rigs -- focal lengths 0.5 .. 1.6 image widths, 0 / 4 / 5 / 8 distortion coefficients of realistic size,
    relative rotations up to 0.08 rad per axis, horizontal rigs of either sign and vertical ones, image sizes 160 x 120 ..
    641 x 479: `calib.stereo_rectify` against `oracle/src/calib.c` to 1e-11 with identical ROIs, the fixed-point maps within
    one LSB on fewer than 1e-3 of the pixels."""
    rng = np.random.default_rng(77)
    for it in range(16):
        w, h = int(rng.choice([160, 320, 641])), int(rng.choice([120, 240, 479]))
        f = rng.uniform(0.5, 1.6) * w
        K1 = np.array([[f, 0, w / 2 + rng.uniform(-20, 20)], [0, f * rng.uniform(0.95, 1.05), h / 2 + rng.uniform(-20, 20)], [0, 0, 1]])
        K2 = K1.copy()
        K2[0, 0] *= rng.uniform(0.95, 1.05); K2[1, 1] *= rng.uniform(0.95, 1.05); K2[0, 2] += rng.uniform(-10, 10); K2[1, 2] += rng.uniform(-10, 10)
        nd = int(rng.choice([0, 4, 5, 8]))

        def dist():
            d = np.array([rng.uniform(-0.4, 0.3), rng.uniform(-0.1, 0.2), rng.uniform(-0.005, 0.005), rng.uniform(-0.005, 0.005),
                          rng.uniform(-0.05, 0.05), rng.uniform(-0.1, 0.1), rng.uniform(-0.05, 0.05), rng.uniform(-0.02, 0.02)])
            return d[:nd] if nd else np.zeros(5)
        d1, d2 = dist(), dist()
        R = calib.rodrigues_vec_to_mat(rng.uniform(-0.08, 0.08, 3))
        B = rng.uniform(0.05, 0.6)
        if rng.random() < 0.25:
            T = np.array([rng.uniform(-0.02, 0.02), -B, rng.uniform(-0.02, 0.02)])
        else:
            T = np.array([-B * rng.choice([1, -1]), rng.uniform(-0.02, 0.02), rng.uniform(-0.02, 0.02)])
        got = calib.stereo_rectify(K1, d1, K2, d2, (w, h), R, T)
        ref = oracle.stereo_rectify(K1, d1, K2, d2, (w, h), R, T)
        for g, r, what in zip(got[:5], ref[:5], ("R1", "R2", "P1", "P2", "Q")):
            assert np.allclose(g, r, rtol=0, atol=1e-11), (it, what, float(np.abs(np.asarray(g) - r).max()))
        assert tuple(got[5]) == tuple(ref[5]) and tuple(got[6]) == tuple(ref[6]), (it, got[5:], ref[5:])
        for K, d, Rk, P in ((K1, d1, got[0], got[2]), (K2, d2, got[1], got[3])):
            m1, m2 = calib.init_undistort_rectify_map(K, d, Rk, P, (w, h))
            o1, o2 = oracle.init_undistort_rectify_map(K, d, Rk, P, (w, h))
            fu, fv = m1[..., 0].astype(np.int64) * 32 + (m2 & 31), m1[..., 1].astype(np.int64) * 32 + (m2 >> 5)
            gu, gv = o1[..., 0].astype(np.int64) * 32 + (o2 & 31), o1[..., 1].astype(np.int64) * 32 + (o2 >> 5)
            du, dv = np.abs(fu - gu), np.abs(fv - gv)
            assert du.max() <= 1 and dv.max() <= 1 and ((du + dv) > 0).mean() < 1e-3, (it, int(du.max()), int(dv.max()))
