"""Analytic known-answer tests pinning the oracle's restatement of the cv2 stages (the reference
holds no fixture for them -- SURVEY.md section 4)."""
import numpy as np

from openvo_amd.synth import Corridor

P = dict(minDisparity=0, numDisparities=32, blockSize=5, P1=200, P2=800, disp12MaxDiff=1, preFilterCap=63,
         uniquenessRatio=10, speckleWindowSize=100, speckleRange=2)


def _texture(h, w, seed=0):
    rng = np.random.default_rng(seed)
    a = rng.integers(0, 256, (h // 4 + 2, w // 4 + 2)).astype(np.float64)
    a = np.kron(a, np.ones((4, 4)))[:h, :w]
    a = (a + np.roll(a, 1, 0) + np.roll(a, 1, 1) + np.roll(a, 2, 1)) / 4
    return a.astype(np.uint8)


def test_sgbm_constant_disparity_plane(oracle):
    L = _texture(96, 200, 1)
    d = 7
    R = np.roll(L, -d, axis=1)           # right image sees the scene shifted left by d
    for mode in (0, 1):
        disp = oracle.sgbm_compute(L, R, P, mode)
        assert disp.dtype == np.int16 and disp.shape == L.shape
        assert (disp[:, :32] == -16).all()                      # columns < numDisparities: INVALID = (minD-1)*16
        inner = disp[8:-8, 48:-16]
        assert (inner == d * 16).mean() > 0.97                  # exact integer disparity, zero sub-pixel offset


def test_sgbm_subpixel_and_float_conversion(oracle):
    c = Corridor("T0")
    L, R = c.pair(0)
    disp = oracle.sgbm_compute(L, R, c.sgbm_params(), 0)
    f = disp.astype(np.float32) / 16                            # reference stereo_camera.py:51, exact in f32
    assert np.array_equal((f * 16).astype(np.int16), disp)
    # ground plane row v: d = B (v - cy) / 1.65
    v = 200
    row = f[v][(f[v] > 0)]
    assert abs(np.median(row) - c.B * (v - c.cy) / 1.65) < 0.25


def test_cost_volume_definition(oracle):
    """C = P2 + 5x5 replicate-border box sum of the BT pixel cost; checked by brute force."""
    rng = np.random.default_rng(3)
    L = rng.integers(0, 256, (12, 48), dtype=np.uint8)
    R = rng.integers(0, 256, (12, 48), dtype=np.uint8)
    p = dict(P, numDisparities=16)
    C = oracle.sgbm_cost_volume(L, R, p).astype(np.int64)
    h, w = L.shape
    D, ft = 16, 63

    def chan(img, c):
        I = img.astype(np.int64)
        if c == 1:
            out = I.copy()
        else:
            Ip = np.pad(I, ((1, 1), (0, 0)), mode="edge")
            out = np.zeros_like(I)
            for y in range(h):
                for x in range(1, w - 1):
                    g = (Ip[y + 1, x + 1] - Ip[y + 1, x - 1]) * 2 + Ip[y, x + 1] - Ip[y, x - 1] + Ip[y + 2, x + 1] - Ip[y + 2, x - 1]
                    out[y, x] = min(max(g, -ft), ft) + ft
        out[:, 0] = out[:, -1] = ft
        return out

    def bounds(a):
        lo, hi = a.copy(), a.copy()
        for x in range(w):
            vl = (a[:, x] + a[:, x - 1]) // 2 if x > 0 else a[:, x]
            vr = (a[:, x] + a[:, x + 1]) // 2 if x < w - 1 else a[:, x]
            lo[:, x] = np.minimum(np.minimum(vl, vr), a[:, x])
            hi[:, x] = np.maximum(np.maximum(vl, vr), a[:, x])
        return lo, hi

    W1 = w - D
    pix = np.zeros((h, W1, D), np.int64)
    for c in (0, 1):
        u, v = chan(L, c), chan(R, c)
        u0, u1 = bounds(u)
        v0, v1 = bounds(v)
        for x1 in range(W1):
            x = x1 + D
            for d in range(D):
                c0 = np.maximum(np.maximum(0, u[:, x] - v1[:, x - d]), v0[:, x - d] - u[:, x])
                c1 = np.maximum(np.maximum(0, v[:, x - d] - u1[:, x]), u0[:, x] - v[:, x - d])
                pix[:, x1, d] += np.minimum(c0, c1) >> (0 if c == 0 else 2)
    pp = np.pad(pix, ((2, 2), (2, 2), (0, 0)), mode="edge")
    box = sum(pp[2 + j: 2 + j + h, 2 + i: 2 + i + W1] for j in range(-2, 3) for i in range(-2, 3))
    assert np.array_equal(C, box + 800)


def test_median_and_speckle(oracle):
    rng = np.random.default_rng(4)
    img = rng.integers(-16, 2000, (20, 30)).astype(np.int16)
    med = oracle.median3x3_s16(img)
    p = np.pad(img, 1, mode="edge")
    ref = np.median(np.stack([p[1 + j: 21 + j, 1 + i: 31 + i] for j in (-1, 0, 1) for i in (-1, 0, 1)]), 0)
    assert np.array_equal(med, ref.astype(np.int16))
    d = np.full((20, 30), 320, np.int16)
    d[5:8, 5:8] = 800          # 9-pixel blob: removed when window >= 9
    d[12, 20] = -16            # already invalid
    assert (oracle.filter_speckles(d, -16, 9, 32)[5:8, 5:8] == -16).all()
    assert (oracle.filter_speckles(d, -16, 8, 32)[5:8, 5:8] == 800).all()
    out = oracle.filter_speckles(d, -16, 9, 32)
    assert (out[d == 320] == 320).all()


def test_fast_corner_and_nms(oracle):
    img = np.full((40, 40), 50, np.uint8)
    img[20, 20] = 200                      # isolated bright pixel: all 16 ring pixels darker
    img[10, 30] = 90                       # contrast 40: corner with a lower score
    sm = oracle.fast_score_map(img, 20)
    ys, xs = np.nonzero(sm)
    assert sorted(zip(ys.tolist(), xs.tolist())) == [(10, 30), (20, 20)]
    assert sm[20, 20] == 149 and sm[10, 30] == 39   # largest threshold keeping it a corner
    img[20, 21] = 200                      # two equal neighbours: strict NMS suppresses both
    assert oracle.fast_score_map(img, 20)[20, 20:22].sum() == 0
    flat = np.full((40, 40), 77, np.uint8)
    assert oracle.fast_score_map(flat, 20).sum() == 0


def test_orb_quotas_canonical_order_and_rotation(oracle):
    c = Corridor("C1")
    L, _ = c.pair(1)
    k = oracle.orb_detect_and_compute(L, None, 500)
    n = len(k["xy"])
    assert 400 <= n <= 520
    cnt = np.bincount(k["octave"], minlength=8)
    assert (cnt <= np.array([109, 90, 75, 63, 52, 44, 36, 31]) + 3).all()
    scale = (1.2 ** k["octave"]).astype(np.float32)
    x, y = k["xy"][:, 0] / scale, k["xy"][:, 1] / scale
    key = k["octave"] * 1e8 + np.rint(y) * 1e4 + np.rint(x)
    assert (np.diff(key) > 0).all()                              # canonical (octave, y, x) order
    assert (k["angle"] >= 0).all() and (k["angle"] < 360).all()
    assert np.allclose(k["size"], 31 * scale)
    # the image rotated by 180 degrees gives angles rotated by 180 degrees for matching corners
    k2 = oracle.orb_detect_and_compute(L[::-1, ::-1].copy(), None, 500)
    lv0 = k["octave"] == 0
    p0 = {(int(a), int(b)): ang for (a, b), ang in zip(k["xy"][lv0], k["angle"][lv0])}
    h, w = L.shape
    hits = 0
    for (a, b), ang in zip(k2["xy"][k2["octave"] == 0], k2["angle"][k2["octave"] == 0]):
        q = (w - 1 - int(a), h - 1 - int(b))
        if q in p0:
            dd = abs(((ang - p0[q]) % 360) - 180)
            assert dd < 1e-3
            hits += 1
    assert hits > 20


def test_orb_mask_excludes_keypoints(oracle):
    c = Corridor("C1")
    L, _ = c.pair(1)
    mask = np.zeros_like(L)
    mask[:, : L.shape[1] // 2] = 255
    k = oracle.orb_detect_and_compute(L, mask, 500)
    assert len(k["xy"]) > 50 and (k["xy"][:, 0] < L.shape[1] // 2 + 1).all()


def test_resize_linear_exact_constant_and_ramp(oracle):
    a = np.full((30, 40), 131, np.uint8)
    assert (oracle.resize_linear_exact(a, 33, 25) == 131).all()
    ramp = np.tile(np.arange(0, 240, 2, dtype=np.uint8), (10, 1))
    out = oracle.resize_linear_exact(ramp, 100, 8).astype(int)
    assert (np.diff(out[0]) >= 0).all() and abs(out[0][50] - (1.2 * 50.5 - 0.5) * 2) <= 1.5


def test_bf_knn2_ties_and_ratio(oracle):
    t = np.zeros((5, 32), np.uint8)
    t[1, 0] = 0b1          # distance 1
    t[2, 0] = 0b10         # distance 1 (tie with row 1 -> lower index first)
    t[3, 0] = 0b111
    t[4] = 255
    q = np.zeros((1, 32), np.uint8)
    idx, dist = oracle.bf_knn2_hamming(q, t)
    assert idx.tolist() == [[0, 1]] and dist.tolist() == [[0, 1]]
    idx, dist = oracle.bf_knn2_hamming(q, t[1:])
    assert idx.tolist() == [[0, 1]] and dist.tolist() == [[1, 1]]
    qo, to = oracle.ratio_filter(np.array([[3, 4], [1, 2]], np.int32), np.array([[8, 10], [7, 10]], np.int32), 0.8)
    assert qo.tolist() == [1] and to.tolist() == [1]            # 8 < 0.8*10 is False (strict), 7 < 8 True


def test_umeyama_exact_rigid_motion_and_reflection(oracle):
    rng = np.random.default_rng(9)
    src = rng.normal(size=(50, 3)).astype(np.float32) * 3
    ang = 0.2
    Rm = np.array([[np.cos(ang), -np.sin(ang), 0], [np.sin(ang), np.cos(ang), 0], [0, 0, 1]])
    t = np.array([0.3, -0.2, 1.0])
    dst = (src.astype(np.float64) @ Rm.T + t).astype(np.float32)
    T, s = oracle.umeyama(src, dst)
    assert np.allclose(T[:, :3], Rm, atol=1e-6) and np.allclose(T[:, 3], t, atol=1e-5) and abs(s - 1) < 1e-6
    assert abs(np.linalg.det(T[:, :3]) - 1) < 1e-12
    # a mirrored cloud must still give a proper rotation (force_rotation=True)
    T2, _ = oracle.umeyama(src, dst * np.array([1, 1, -1], np.float32))
    assert abs(np.linalg.det(T2[:, :3]) - 1) < 1e-9
    r = oracle.rodrigues(T[:, :3])
    assert abs(np.linalg.norm(r) - ang) < 1e-6 and abs(r[2, 0] - ang) < 1e-6
    assert np.allclose(oracle.rodrigues(np.eye(3)), 0)
    U, w, Vt = oracle.svd3(rng.normal(size=(3, 3)))
    assert w[0] >= w[1] >= w[2] >= 0 and np.allclose(U @ U.T, np.eye(3), atol=1e-12)


def test_reproject_matches_definition(oracle):
    c = Corridor("T0")
    Q = c.Q()
    disp = np.array([[16.0, 0.0, -1.0, 5.5]], np.float32)
    with np.errstate(all="ignore"):
        out = oracle.reproject_to_3d(disp, Q)
    z0 = c.f * c.B / 16.0
    assert abs(out[0, 0, 2] - z0) < 1e-4
    assert np.isinf(out[0, 1, 2])                                # d = 0 -> W = 0 -> inf
    assert out[0, 2, 2] < 0                                      # invalid disparity -1 gives finite garbage
    pts, st = oracle.points3d_at(np.array([[256, 0, -16, 88]], np.int16), Q, (0, 0, 4, 1), np.array([[0.0, 0.0], [3.0, 0.0]], np.float32))
    assert st.tolist() == [0, 0] and np.array_equal(pts[0], out[0, 0]) and np.array_equal(pts[1], out[0, 3])


def test_bgr2gray_and_remap(oracle):
    bgr = np.zeros((1, 4, 3), np.uint8)
    bgr[0, 0] = (255, 255, 255)
    bgr[0, 1] = (255, 0, 0)
    bgr[0, 2] = (0, 255, 0)
    bgr[0, 3] = (0, 0, 255)
    assert oracle.bgr2gray(bgr).tolist() == [[255, 29, 150, 76]]
    src = np.arange(64, dtype=np.uint8).reshape(8, 8) * 3
    m1 = np.zeros((8, 8, 2), np.int16)
    m1[..., 0], m1[..., 1] = np.meshgrid(np.arange(8), np.arange(8))
    m2 = np.zeros((8, 8), np.uint16)
    assert np.array_equal(oracle.remap_bilinear(src, m1, m2), src)            # identity map
    m2[:] = 16                                                               # fx = 16/32: halfway to x+1
    out = oracle.remap_bilinear(src, m1, m2)
    assert out[0, 0] == (int(src[0, 0]) + int(src[0, 1]) + 1) // 2
    assert out[0, 7] == (int(src[0, 7]) * 16384 + (1 << 14)) >> 15           # x+1 outside: border 0


def test_p3p_recovers_synthetic_poses(oracle):
    """The PnP loop has no reference counterpart; its restatement is pinned by geometry instead: for
    random poses and triangles the true pose is among the P3P candidates (float64, 1e-7)."""
    rng = np.random.default_rng(0)
    for trial in range(300):
        r = rng.normal(size=3) * 0.3
        th = np.linalg.norm(r)
        k = r / th
        Kx = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
        R = np.eye(3) + np.sin(th) * Kx + (1 - np.cos(th)) * Kx @ Kx
        t = rng.normal(size=3) * 0.5
        X = rng.uniform(-2, 2, (3, 3)) + np.array([0, 0, 6.0])
        Xc = X @ R.T + t
        y = Xc / np.linalg.norm(Xc, axis=1, keepdims=True)
        sols = oracle.p3p(y, X)
        assert 1 <= len(sols) <= 4
        assert min(np.abs(Rs - R).max() + np.abs(ts - t).max() for Rs, ts in sols) < 1e-7
        for Rs, ts in sols:   # every candidate is a rotation that reproduces the three bearings
            assert np.allclose(Rs @ Rs.T, np.eye(3), atol=1e-8) and np.linalg.det(Rs) > 0
            Yc = X @ Rs.T + ts
            assert np.allclose(Yc / np.linalg.norm(Yc, axis=1, keepdims=True), y, atol=1e-7)


def test_ransac_pnp_oracle_finds_the_pose(oracle):
    rng = np.random.default_rng(3)
    n, f, cx, cy = 400, 700.0, 320.0, 240.0
    X = np.stack([rng.uniform(-6, 6, n), rng.uniform(-3, 3, n), rng.uniform(4, 30, n)], 1)
    R = np.eye(3)
    t = np.array([0.1, 0.0, -0.25])
    Xc = X @ R.T + t
    uv = np.stack([f * Xc[:, 0] / Xc[:, 2] + cx, f * Xc[:, 1] / Xc[:, 2] + cy], 1)
    out = rng.choice(n, 120, replace=False)
    uv[out] += rng.uniform(-50, 50, (120, 2))
    r = oracle.ransac_pnp(X, uv, [f, f, cx, cy], iters=200, thr=1.0, seed=7)
    assert r["best_count"] >= 280 and r["counts"].max() == r["best_count"] and r["counts"][r["best_iter"]] == r["best_count"]
    assert np.abs(r["Rt"][:, :3] - R).max() < 1e-3 and np.abs(r["Rt"][:, 3] - t).max() < 1e-2
    assert r["mask"].sum() == r["best_count"]
