"""One-off rectification setup of StereoCamera.__init__ without cv2 (numpy, host side).

Restates cv2.stereoRectify (default flags CALIB_ZERO_DISPARITY, alpha = -1) and
cv2.initUndistortRectifyMap(..., CV_16SC2) [reference stereo_camera.py:17-22] from OpenCV 4.x
calib3d/src/calibration.cpp (cvStereoRectify, icvGetRectangles) and
imgproc/src/undistort.dispatch.cpp.  Setup runs once per camera, so it stays on the host
(SURVEY.md section 8(f) row 1).  Distortion model: k1 k2 p1 p2 k3 [k4 k5 k6 [s1 s2 s3 s4]].
"""
import numpy as np

INTER_BITS = 5
INTER_TAB_SIZE = 1 << INTER_BITS


def rodrigues_vec_to_mat(om):
    om = np.asarray(om, np.float64).reshape(3)
    theta = np.linalg.norm(om)
    if theta < 2.220446049250313e-16:
        return np.eye(3)
    k = om / theta
    K = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
    return np.cos(theta) * np.eye(3) + (1 - np.cos(theta)) * np.outer(k, k) + np.sin(theta) * K


def rodrigues_mat_to_vec(R):
    R = np.asarray(R, np.float64)
    U, _, Vt = np.linalg.svd(R)
    R = U @ Vt
    r = np.array([R[2, 1] - R[1, 2], R[0, 2] - R[2, 0], R[1, 0] - R[0, 1]])
    s = np.sqrt((r @ r) * 0.25)
    c = np.clip((np.trace(R) - 1) * 0.5, -1.0, 1.0)
    theta = np.arccos(c)
    if s < 1e-5:
        if c > 0:
            return np.zeros(3)
        t = np.sqrt(np.maximum((np.diag(R) + 1) * 0.5, 0))
        t[1] *= -1.0 if R[0, 1] < 0 else 1.0
        t[2] *= -1.0 if R[0, 2] < 0 else 1.0
        if abs(t[0]) < abs(t[1]) and abs(t[0]) < abs(t[2]) and (R[1, 2] > 0) != (t[1] * t[2] > 0):
            t[2] = -t[2]
        return t * (theta / np.linalg.norm(t))
    return r * (theta / (2 * s))


def _dist14(dist):
    d = np.zeros(14, np.float64)
    if dist is not None:
        v = np.asarray(dist, np.float64).ravel()
        d[:min(len(v), 14)] = v[:14]
    return d


def undistort_points(pts, K, dist, R=None, P=None, iters=5):
    """cv2.undistortPoints: pixel -> (optionally rectified / re-projected) coordinates."""
    pts = np.asarray(pts, np.float64).reshape(-1, 2)
    K = np.asarray(K, np.float64)
    k = _dist14(dist)
    fx, fy, cx, cy = K[0, 0], K[1, 1], K[0, 2], K[1, 2]
    x0 = (pts[:, 0] - cx) / fx
    y0 = (pts[:, 1] - cy) / fy
    x, y = x0.copy(), y0.copy()
    if np.any(k != 0):
        for _ in range(iters):
            r2 = x * x + y * y
            icdist = (1 + ((k[7] * r2 + k[6]) * r2 + k[5]) * r2) / (1 + ((k[4] * r2 + k[1]) * r2 + k[0]) * r2)
            icdist = np.where(icdist < 0, 1.0, icdist)
            dx = 2 * k[2] * x * y + k[3] * (r2 + 2 * x * x) + k[8] * r2 + k[9] * r2 * r2
            dy = k[2] * (r2 + 2 * y * y) + 2 * k[3] * x * y + k[10] * r2 + k[11] * r2 * r2
            x = (x0 - dx) * icdist
            y = (y0 - dy) * icdist
    if R is not None:
        R = np.asarray(R, np.float64)
        X = np.stack([x, y, np.ones_like(x)], 1) @ R.T
        x, y = X[:, 0] / X[:, 2], X[:, 1] / X[:, 2]
    if P is not None:
        P = np.asarray(P, np.float64)
        x, y = x * P[0, 0] + P[0, 2], y * P[1, 1] + P[1, 2]
    return np.stack([x, y], 1)


def _get_rectangles(K, dist, R, P, img_size):
    """icvGetRectangles: inscribed / circumscribed rectangles of the rectified 9x9 grid."""
    N = 9
    w, h = img_size
    gx, gy = np.meshgrid(np.arange(N) * (w - 1) / (N - 1), np.arange(N) * (h - 1) / (N - 1))
    grid = np.stack([gx.ravel(), gy.ravel()], 1).astype(np.float32)
    p = undistort_points(grid, K, dist, R, P).astype(np.float32).astype(np.float64).reshape(N, N, 2)
    p32 = p.astype(np.float32)       # cv::Rect_<float>: extents are float32 differences
    ix0, ix1 = p32[:, 0, 0].max(), p32[:, N - 1, 0].min()
    iy0, iy1 = p32[0, :, 1].max(), p32[N - 1, :, 1].min()
    inner = (float(ix0), float(iy0), float(np.float32(ix1 - ix0)), float(np.float32(iy1 - iy0)))
    ox0, ox1, oy0, oy1 = p32[..., 0].min(), p32[..., 0].max(), p32[..., 1].min(), p32[..., 1].max()
    outer = (float(ox0), float(oy0), float(np.float32(ox1 - ox0)), float(np.float32(oy1 - oy0)))
    return inner, outer


def stereo_rectify(K1, D1, K2, D2, img_size, R, T):
    """cv2.stereoRectify(K1, D1, K2, D2, imageSize, R, T) with its Python defaults.

    Returns R1, R2, P1, P2, Q, roi1, roi2 (rois are (x, y, w, h) tuples of ints)."""
    K1, K2 = np.asarray(K1, np.float64), np.asarray(K2, np.float64)
    R = np.asarray(R, np.float64).reshape(3, 3)
    T = np.asarray(T, np.float64).reshape(3)
    nx, ny = int(img_size[0]), int(img_size[1])
    om = rodrigues_mat_to_vec(R) * -0.5
    r_r = rodrigues_vec_to_mat(om)          # rotate both cameras half way
    t = r_r @ T
    idx = 0 if abs(t[0]) > abs(t[1]) else 1
    c, nt = t[idx], np.linalg.norm(t)
    uu = np.zeros(3)
    uu[idx] = 1.0 if c > 0 else -1.0
    ww = np.cross(t, uu)
    nw = np.linalg.norm(ww)
    if nw > 0.0:
        ww = ww * (np.arccos(abs(c) / nt) / nw)
    wR = rodrigues_vec_to_mat(ww)
    R1 = wR @ r_r.T
    R2 = wR @ r_r
    t = R2 @ T
    # OpenCV >= 3.4.8 / 4.1.2 (the reference needs >= 4.5.5 for estimateAffine3D(force_rotation)):
    # fc_new = (K1[idx^1, idx^1] + K2[idx^1, idx^1]) * ratio, ratio = newImageSize / imageSize / 2 = 0.5
    # when newImageSize is left at its default; no k1-dependent shrink (that rule is pre-3.4.8)
    fc_new = (K1[idx ^ 1, idx ^ 1] + K2[idx ^ 1, idx ^ 1]) * 0.5
    cc = []
    # cvStereoRectify keeps the four corners in CvPoint2D32f: undistorted points, their homogeneous
    # form and the projected points are float32; the average is taken in double
    corners = np.array([[0, 0], [nx - 1, 0], [0, ny - 1], [nx - 1, ny - 1]], np.float32)
    for K, D, Rk in ((K1, D1, R1), (K2, D2, R2)):
        n = undistort_points(corners, K, D).astype(np.float32).astype(np.float64)
        X = np.stack([n[:, 0], n[:, 1], np.ones(4)], 1) @ Rk.T
        proj = np.stack([fc_new * X[:, 0] / X[:, 2], fc_new * X[:, 1] / X[:, 2]], 1).astype(np.float32)
        avg = proj.astype(np.float64).mean(0)
        cc.append(np.array([(nx - 1) / 2 - avg[0], (ny - 1) / 2 - avg[1]]))
    # CALIB_ZERO_DISPARITY: both principal points become their average
    cc[0] = cc[1] = (cc[0] + cc[1]) * 0.5
    P1 = np.zeros((3, 4))
    P2 = np.zeros((3, 4))
    for P, c_ in ((P1, cc[0]), (P2, cc[1])):
        P[0, 0] = P[1, 1] = fc_new
        P[0, 2], P[1, 2], P[2, 2] = c_[0], c_[1], 1.0
    P2[idx, 3] = t[idx] * fc_new
    inner1, _ = _get_rectangles(K1, D1, R1, P1, (nx, ny))
    inner2, _ = _get_rectangles(K2, D2, R2, P2, (nx, ny))

    def roi(inner, cx0, cy0):
        x, y = int(np.ceil(inner[0] - cx0 + cx0)), int(np.ceil(inner[1] - cy0 + cy0))
        w, h = int(np.floor(inner[2])), int(np.floor(inner[3]))
        x0, y0 = max(x, 0), max(y, 0)
        x1, y1 = min(x + w, nx), min(y + h, ny)
        if x1 <= x0 or y1 <= y0:
            return (0, 0, 0, 0)
        return (x0, y0, x1 - x0, y1 - y0)
    roi1 = roi(inner1, cc[0][0], cc[0][1])
    roi2 = roi(inner2, cc[1][0], cc[1][1])
    Q = np.array([[1, 0, 0, -cc[0][0]], [0, 1, 0, -cc[0][1]], [0, 0, 0, fc_new],
                  [0, 0, -1.0 / t[idx], (cc[0][0] - cc[1][0]) / t[idx] if idx == 0 else (cc[0][1] - cc[1][1]) / t[idx]]],
                 np.float64)
    return R1, R2, P1, P2, Q, roi1, roi2


def init_undistort_rectify_map(K, dist, R, P, img_size):
    """cv2.initUndistortRectifyMap(K, dist, R, P, size, CV_16SC2) -> (map1 int16 HxWx2, map2 uint16 HxW)."""
    K = np.asarray(K, np.float64)
    k = _dist14(dist)
    w, h = int(img_size[0]), int(img_size[1])
    R = np.eye(3) if R is None else np.asarray(R, np.float64)
    A = np.asarray(P, np.float64)[:3, :3]
    ir = np.linalg.inv(A @ R)
    j = np.arange(w, dtype=np.float64)[None, :]
    i = np.arange(h, dtype=np.float64)[:, None]
    _x = i * ir[0, 1] + ir[0, 2] + j * ir[0, 0]
    _y = i * ir[1, 1] + ir[1, 2] + j * ir[1, 0]
    _w = i * ir[2, 1] + ir[2, 2] + j * ir[2, 0]
    x, y = _x / _w, _y / _w
    x2, y2 = x * x, y * y
    r2 = x2 + y2
    _2xy = 2 * x * y
    kr = (1 + ((k[4] * r2 + k[1]) * r2 + k[0]) * r2) / (1 + ((k[7] * r2 + k[6]) * r2 + k[5]) * r2)
    xd = x * kr + k[2] * _2xy + k[3] * (r2 + 2 * x2) + k[8] * r2 + k[9] * r2 * r2
    yd = y * kr + k[2] * (r2 + 2 * y2) + k[3] * _2xy + k[10] * r2 + k[11] * r2 * r2
    u = K[0, 0] * xd + K[0, 2]
    v = K[1, 1] * yd + K[1, 2]
    lim = 2.0 ** 31 - 1
    iu = np.rint(np.clip(u * INTER_TAB_SIZE, -lim, lim)).astype(np.int64)
    iv = np.rint(np.clip(v * INTER_TAB_SIZE, -lim, lim)).astype(np.int64)
    map1 = np.empty((h, w, 2), np.int16)
    map1[..., 0] = (iu >> INTER_BITS).astype(np.int16)   # (short) cast wraps like C
    map1[..., 1] = (iv >> INTER_BITS).astype(np.int16)
    map2 = ((iv & (INTER_TAB_SIZE - 1)) * INTER_TAB_SIZE + (iu & (INTER_TAB_SIZE - 1))).astype(np.uint16)
    return map1, map2
