"""GPU parity: every HIP stage, called through the C ABI, against the CPU oracle on the same
seeded inputs.  Integer/byte/index stages must be bit-exact; float stages state their tolerance."""
import os

import numpy as np
import pytest

from openvo_amd.synth import Corridor

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")


def _pair(name="T0", k=3):
    c = Corridor(name)
    L, R = c.pair(k)
    return c, L, R


@pytest.mark.parametrize("cfg,mode", [("T0", 0), ("T0", 1), ("C1", 0)])
def test_sgbm_bit_exact(oracle, ctx_small, cfg, mode):
    c, L, R = _pair(cfg)
    p = c.sgbm_params()
    ctx_small.set_sgbm(p, mode)
    got = ctx_small.sgbm_compute_host(L, R)
    ref = oracle.sgbm_compute(L, R, p, mode)
    assert got.shape == ref.shape
    nbad = int((got != ref).sum())
    assert nbad == 0, "%d of %d disparity pixels differ" % (nbad, ref.size)
    assert (ref >= 0).mean() > 0.5  # the scene really produced disparities


@pytest.mark.parametrize("params", [
    dict(minDisparity=0, numDisparities=48, blockSize=5, P1=8, P2=32, disp12MaxDiff=1, preFilterCap=31,
         uniquenessRatio=15, speckleWindowSize=0, speckleRange=0),
    dict(minDisparity=-16, numDisparities=32, blockSize=5, P1=200, P2=800, disp12MaxDiff=2, preFilterCap=63,
         uniquenessRatio=5, speckleWindowSize=50, speckleRange=1),
    dict(minDisparity=4, numDisparities=16, blockSize=3, P1=24, P2=96, disp12MaxDiff=1, preFilterCap=63,
         uniquenessRatio=10, speckleWindowSize=100, speckleRange=2),
    dict(minDisparity=0, numDisparities=64, blockSize=9, P1=100, P2=1000, disp12MaxDiff=-1, preFilterCap=10,
         uniquenessRatio=0, speckleWindowSize=20, speckleRange=4),
])
def test_sgbm_parameter_corners(oracle, ctx_small, params):
    c, L, R = _pair("T0", 7)
    ctx_small.set_sgbm(params, 0)
    got = ctx_small.sgbm_compute_host(L, R)
    ref = oracle.sgbm_compute(L, R, params, 0)
    assert np.array_equal(got, ref)


@pytest.fixture(scope="module")
def ctx_wide():
    from openvo_amd import _native
    c = _native.Context(0, 640, 256, 256, 1000)
    yield c
    c.close()


@pytest.mark.parametrize("ndisp,ur,mode", [(160, 10, 0), (192, 15, 1), (256, 10, 0), (240, 5, 1), (96, 99, 0), (64, 100, 0), (128, 0, 1)])
def test_sgbm_wide_disparity_ranges(oracle, ctx_wide, ndisp, ur, mode):
    """Every register-count variant of the aggregation/WTA kernels (D/32 = 2..8, padded and exact),
    both path sets, and both forms of the uniqueness test (threshold form < 100 <= product form)."""
    c, L, R = _pair("C1", 2)
    L, R = L[:200], R[:200]
    p = dict(minDisparity=0, numDisparities=ndisp, blockSize=5, P1=200, P2=800, disp12MaxDiff=1, preFilterCap=63,
             uniquenessRatio=ur, speckleWindowSize=0, speckleRange=0)
    ctx_wide.set_sgbm(p, mode)
    got = ctx_wide.sgbm_compute_host(L, R)
    ref = oracle.sgbm_compute(L, R, p, mode)
    assert np.array_equal(got, ref), "%d pixels differ" % int((got != ref).sum())


@pytest.mark.parametrize("block,mind,ndisp", [(1, 0, 64), (3, -16, 48), (7, 0, 128), (7, 5, 32), (11, -16, 160), (11, 0, 16)])
def test_sgbm_cost_volume_block_sizes_interior_and_border_strips(oracle, ctx_wide, block, mind, ndisp):
    """The cost kernel stages the right-image planes of interior column strips in LDS and keeps lane-shift chains for the strips
    at the image border (and wherever a lane's position would leave the row): every window size it is instantiated for, with
    positive, zero and negative minDisparity, one and two waves of disparities -- the disparity stays bit-exact."""
    c, L, R = _pair("C1", 3)
    L, R = L[:120], R[:120]
    p = dict(minDisparity=mind, numDisparities=ndisp, blockSize=block, P1=8 * block * block, P2=32 * block * block, disp12MaxDiff=1,
             preFilterCap=63, uniquenessRatio=10, speckleWindowSize=0, speckleRange=0)
    ctx_wide.set_sgbm(p, 0)
    got = ctx_wide.sgbm_compute_host(L, R)
    ref = oracle.sgbm_compute(L, R, p, 0)
    assert np.array_equal(got, ref), "%d pixels differ" % int((got != ref).sum())


def test_sgbm_random_noise_images(oracle, ctx_small):
    rng = np.random.default_rng(5)
    L = rng.integers(0, 256, (96, 160), dtype=np.uint8)
    R = np.roll(L, -5, axis=1)
    p = dict(minDisparity=0, numDisparities=32, blockSize=5, P1=200, P2=800, disp12MaxDiff=1, preFilterCap=63,
             uniquenessRatio=10, speckleWindowSize=100, speckleRange=2)
    ctx_small.set_sgbm(p, 0)
    assert np.array_equal(ctx_small.sgbm_compute_host(L, R), oracle.sgbm_compute(L, R, p, 0))
    Z = np.zeros_like(L)  # constant images: every cost ties
    assert np.array_equal(ctx_small.sgbm_compute_host(Z, Z), oracle.sgbm_compute(Z, Z, p, 0))


def test_bf_knn2_tile_and_slice_edges_and_adversarial_bit_patterns(oracle):
    """The matcher computes popcount(q ^ t) as |q| + |t| - 2 q.t on the matrix cores (int8 contraction of the bits), walks the
    train set in tiles of 16 descriptors, cuts it into up to 16 slices merged through a ticket, and serves 8 groups of 64
    queries per workgroup: sizes around every one of those boundaries (incl. a train set larger than one LDS chunk of 512 x 16
    slices), all-zero / all-one / single-bit descriptors (norms 0 and 256, dot products 0 and 256), duplicated train
    descriptors (ties -> lower index) and identical query / train sets -- every index and distance equal to the oracle's."""
    from openvo_amd import _native
    ctx = _native.Context(0, 640, 480, 64, 9000)
    rng = np.random.default_rng(77)
    sizes = [(64, 16), (65, 17), (63, 15), (512, 512), (513, 511), (130, 33), (3, 1), (5, 2), (700, 17), (1, 4097), (2, 8200),
             (1100, 2050), (8012, 8030), (576, 9000)]
    for nq, nt in sizes:
        q = rng.integers(0, 256, (nq, 32), dtype=np.uint8)
        t = rng.integers(0, 256, (nt, 32), dtype=np.uint8)
        q[0] = 0; t[0] = 255                                  # |q| = 0 against |t| = 256: distance 256, the largest
        if nq > 2:
            q[1] = 255; q[2] = 0; q[2, 17] = 0x10             # all ones; a single bit
        if nt > 9:
            t[7] = t[2]; t[9] = 0                             # a duplicate (tie -> index 2 first); all zeros
            t[nt - 1] = t[nt - 2]                             # a tie across the last tile's end
        gi, gd = ctx.bf_knn2(q, t)
        ri, rd = oracle.bf_knn2_hamming(q, t)
        assert np.array_equal(gi, ri) and np.array_equal(gd, rd), (nq, nt)
    d = rng.integers(0, 256, (700, 32), dtype=np.uint8)      # a set against itself: best = itself at distance 0
    gi, gd = ctx.bf_knn2(d, d)
    ri, rd = oracle.bf_knn2_hamming(d, d)
    assert np.array_equal(gi, ri) and np.array_equal(gd, rd) and (gd[:, 0] == 0).all()
    ctx.close()


def test_bf_knn2_slice_hand_off_under_uneven_load(oracle):
    """The slices of a train set are merged inside the launch: write-through stores, the storing wave's wait, one agent-scope
    ticket, write-through-aware loads by the wave whose ticket came last -- no fence.  A hand-off like that has to be tested
    under UNEVEN load with every word checked (MI355X_MICROARCH.md): 60 launches of changing size, each result compared with
    the oracle's in full, while look-ahead engines run disparity + ORB of staged pairs beside them on the same GPU (other
    workgroups occupy CUs and L2s, the matcher's workgroups start at different times and on different XCDs), the scratch
    behind the distances being reused by every launch."""
    from openvo_amd import StereoCamera, StereoOdometer
    from openvo_amd.synth import Corridor
    c = Corridor("C1")
    cam = StereoCamera(c.K(), c.dist(), c.K(), c.dist(), c.rect_params(), c.sgbm_params(), (c.w, c.h), max_keypoints=2000)
    odo = StereoOdometer(cam, nfeatures=500, preprocessed_frames=True)
    staged = cam.stage_pairs(c.pairs(0, 24))
    ctx = cam._ctx
    rng = np.random.default_rng(5)
    base = rng.integers(0, 256, (4000, 32), dtype=np.uint8)
    k = 0
    for rep in range(60):
        if rep % 3 == 0:                                     # (keeps the engines busy: the next pairs' disparity + keypoints start here)
            odo.update(staged[k % 24], None)
            k += 1
        nq, nt = int(rng.integers(65, 2000)), int(rng.integers(40, 4000))
        q = base[rng.integers(0, 4000, nq)] ^ (rng.integers(0, 256, (nq, 32), dtype=np.uint8) & rng.integers(0, 256, (nq, 32), dtype=np.uint8) & 0x11)
        t = base[rng.permutation(4000)[:nt]]                  # near-duplicates of the queries: small distances, many close calls
        gi, gd = ctx.bf_knn2(q, t)
        ri, rd = oracle.bf_knn2_hamming(q, t)
        assert np.array_equal(gi, ri) and np.array_equal(gd, rd), (rep, nq, nt)
    assert ctx.sgbm_sweep_status() == 0
    ctx.close()


@pytest.mark.parametrize("nq,nt", [(500, 500), (1, 2), (7, 1), (3, 0), (1000, 777)])
def test_bf_knn2_bit_exact(oracle, ctx_small, nq, nt):
    rng = np.random.default_rng(nq * 1000 + nt)
    q = rng.integers(0, 256, (nq, 32), dtype=np.uint8)
    t = rng.integers(0, 256, (nt, 32), dtype=np.uint8)
    if nt > 10:
        t[5] = t[3]          # forced distance ties -> lower train index first
        q[0] = t[3]
    gi, gd = ctx_small.bf_knn2(q, t)
    ri, rd = oracle.bf_knn2_hamming(q, t)
    assert np.array_equal(gi, ri) and np.array_equal(gd, rd)


def test_orb_matches_oracle(oracle, ctx_small):
    c, L, R = _pair("C1", 2)
    p = c.sgbm_params()
    disp = oracle.sgbm_compute(L, R, p, 0)
    d = disp.astype(np.float32) / 16
    mask = ((d >= 4) & (d <= 100)).astype(np.uint8) * 255
    for m in (mask, None):
        got = ctx_small.orb_host(L, m, 500)
        ref = oracle.orb_detect_and_compute(L, m, 500)
        assert len(ref["xy"]) > 100
        assert np.array_equal(got["octave"], ref["octave"])
        assert np.array_equal(got["xy"].view(np.uint32), ref["xy"].view(np.uint32))      # keypoint indices
        assert np.array_equal(got["response"].view(np.uint32), ref["response"].view(np.uint32))
        assert np.array_equal(got["size"], ref["size"])
        assert np.array_equal(got["angle"].view(np.uint32), ref["angle"].view(np.uint32))
        assert np.array_equal(got["desc"], ref["desc"])


def test_points3d_and_bilinear(oracle, ctx_small):
    g = np.load(os.path.join(GOLD, "g2_bilinear.npz"))
    out, st = ctx_small.bilinear_at(g["img"], g["xy"])
    ref, rst = oracle.bilinear_at(g["img"], g["xy"])
    assert np.array_equal(st, rst)
    assert np.array_equal(np.isnan(out), np.isnan(ref))
    ok = ~np.isnan(ref)
    assert np.array_equal(out[ok].view(np.uint32), ref[ok].view(np.uint32))
    # and against the reference's own outputs
    good = g["kind"] == 0
    fin = ~np.isnan(g["out"][good])
    assert np.array_equal(out[good][fin].view(np.uint32), g["out"][good][fin].view(np.uint32))
    assert np.array_equal(st == 2, g["kind"] == 2)


def test_umeyama_and_rodrigues(oracle, ctx_small):
    rng = np.random.default_rng(11)
    src = (rng.uniform(-5, 5, (200, 3)) + [0, 0, 10]).astype(np.float32)
    ang = 0.03
    Rm = np.array([[np.cos(ang), 0, np.sin(ang)], [0, 1, 0], [-np.sin(ang), 0, np.cos(ang)]])
    dst = (src @ Rm.T + [0.01, -0.02, -0.25] + rng.normal(scale=0.01, size=src.shape)).astype(np.float32)
    T, s = ctx_small.umeyama(src, dst)
    Tr, sr = oracle.umeyama(src, dst)
    assert np.allclose(T, Tr, rtol=0, atol=1e-12) and abs(s - sr) < 1e-12
    assert np.allclose(T[:, :3], Rm, atol=2e-3)
    assert np.allclose(ctx_small.rodrigues(T[:, :3]), oracle.rodrigues(T[:, :3]), atol=1e-14)


def test_rigid_clique_matches_reference_golden(ctx_small):
    g3 = np.load(os.path.join(GOLD, "g3_rigid.npz"))
    for k in g3.files:
        if k.endswith("_mask"):
            b = k[:-5]
            m = ctx_small.rigid_clique(g3[b + "_prev"], g3[b + "_cur"], float(g3[b + "_thr"]))
            assert np.array_equal(m, g3[k]), b


def _two_view(n, seed, outlier_frac=0.25):
    rng = np.random.default_rng(seed)
    f, cx, cy = 1050.0, 960.0, 540.0
    X = np.c_[rng.uniform(-8, 8, n), rng.uniform(-4, 4, n), rng.uniform(6, 40, n)]
    ang = 0.03
    R = np.array([[np.cos(ang), 0, np.sin(ang)], [0, 1, 0], [-np.sin(ang), 0, np.cos(ang)]])
    t = np.array([0.05, 0.01, -0.25])
    X2 = X @ R.T + t
    p1 = np.c_[f * X[:, 0] / X[:, 2] + cx, f * X[:, 1] / X[:, 2] + cy] + rng.normal(scale=0.3, size=(n, 2))
    p2 = np.c_[f * X2[:, 0] / X2[:, 2] + cx, f * X2[:, 1] / X2[:, 2] + cy] + rng.normal(scale=0.3, size=(n, 2))
    out = rng.choice(n, int(n * outlier_frac), replace=False)
    p2[out] += rng.uniform(-80, 80, size=(len(out), 2))
    return p1.astype(np.float32), p2.astype(np.float32), [f, f, cx, cy], R, t, out


@pytest.mark.parametrize("n,iters", [(2000, 500), (8, 16), (8000, 300), (77, 1000)])
def test_ransac_essential_bit_exact(oracle, ctx_small, n, iters):
    """No openVO counterpart: GPU vs the build's own CPU restatement -- every hypothesis' inlier
    count, the winner, its mask and E identical (E to 1e-12: same operations, both IEEE)."""
    p1, p2, K4, R, t, out = _two_view(n, n + iters)
    ref = oracle.ransac_essential(p1, p2, K4, iters, 1.0, 4321)
    got = ctx_small.ransac_essential(p1, p2, K4, iters, 1.0, 4321, want_counts=True)
    assert np.array_equal(got["counts"], ref["counts"])
    assert got["best_iter"] == ref["best_iter"] and got["best_count"] == ref["best_count"]
    assert np.array_equal(got["mask"], ref["mask"])
    assert np.allclose(got["E"], ref["E"], rtol=0, atol=1e-12)
    if n >= 2000:
        tx = np.array([[0, -t[2], t[1]], [t[2], 0, -t[0]], [-t[1], t[0], 0]])
        Et = tx @ R
        Et /= np.linalg.norm(Et)
        E = got["E"] / np.linalg.norm(got["E"])
        assert min(np.abs(E - Et).max(), np.abs(E + Et).max()) < 0.06   # sanity of the estimator, not a parity bar
        inl = np.setdiff1d(np.arange(n), out)
        assert got["mask"][inl].mean() > 0.95 and got["mask"][out].mean() < 0.1


@pytest.mark.parametrize("n,iters", [(2000, 500), (6, 16), (8000, 300), (77, 1000), (9, 4096)])
def test_ransac_five_point_bit_exact(oracle, ctx_small, n, iters):
    """Five-point solver (no openVO counterpart): one lane per hypothesis on the GPU vs the CPU restatement -- every
    hypothesis' inlier count (so every root found, every pick made identically), the winner, its mask and E."""
    p1, p2, K4, R, t, out = _two_view(n, 3 * n + iters)
    ref = oracle.ransac_essential(p1, p2, K4, iters, 1.0, 4321, solver=5)
    got = ctx_small.ransac_essential(p1, p2, K4, iters, 1.0, 4321, want_counts=True, solver=5)
    assert np.array_equal(got["counts"], ref["counts"])
    assert got["best_iter"] == ref["best_iter"] and got["best_count"] == ref["best_count"]
    assert np.array_equal(got["mask"], ref["mask"])
    assert np.allclose(got["E"], ref["E"], rtol=0, atol=1e-12)
    if n >= 2000:
        tx = np.array([[0, -t[2], t[1]], [t[2], 0, -t[0]], [-t[1], t[0], 0]])
        Et = tx @ R
        Et /= np.linalg.norm(Et)
        E = got["E"] / np.linalg.norm(got["E"])
        assert min(np.abs(E - Et).max(), np.abs(E + Et).max()) < 0.06
        inl = np.setdiff1d(np.arange(n), out)
        assert got["mask"][inl].mean() > 0.95 and got["mask"][out].mean() < 0.1


def test_ransac_five_point_argument_checks(ctx_small):
    from openvo_amd._native import VoError
    p = np.zeros((5, 2), np.float32)
    with pytest.raises(VoError):
        ctx_small.ransac_essential(p, p, [500, 500, 320, 240], 10, 1.0, 1, solver=5)      # needs 6
    with pytest.raises(ValueError):
        ctx_small.ransac_essential(p, p, [500, 500, 320, 240], 10, 1.0, 1, solver=7)


def _pnp_scene(n, seed, outlier_frac=0.3, noise=0.3):
    rng = np.random.default_rng(seed)
    f, cx, cy = 718.856, 640.0, 360.0
    X = np.stack([rng.uniform(-8, 8, n), rng.uniform(-3, 3, n), rng.uniform(4, 40, n)], 1)
    r = np.array([0.01, -0.03, 0.005])
    th = np.linalg.norm(r)
    k = r / th
    Kx = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
    R = np.eye(3) + np.sin(th) * Kx + (1 - np.cos(th)) * Kx @ Kx
    t = np.array([0.05, -0.02, -0.3])
    Xc = X @ R.T + t
    uv = np.stack([f * Xc[:, 0] / Xc[:, 2] + cx, f * Xc[:, 1] / Xc[:, 2] + cy], 1) + rng.normal(0, noise, (n, 2))
    out = rng.choice(n, int(n * outlier_frac), replace=False)
    uv[out] += rng.uniform(-60, 60, size=(len(out), 2))
    return X.astype(np.float32), uv.astype(np.float32), [f, f, cx, cy], R, t, out


@pytest.mark.parametrize("n,iters", [(500, 400), (4, 8), (8000, 300), (61, 1000)])
def test_ransac_pnp_bit_exact(oracle, ctx_small, n, iters):
    """No openVO counterpart: GPU vs the build's own CPU restatement -- every hypothesis' inlier count,
    the winner and its mask identical, the pose to 1e-12 (same IEEE operations in the same order)."""
    X, uv, K4, R, t, out = _pnp_scene(n, 100 + n + iters)
    ref = oracle.ransac_pnp(X, uv, K4, iters, 2.0, 4321)
    got = ctx_small.ransac_pnp(X, uv, K4, iters, 2.0, 4321, want_counts=True)
    assert np.array_equal(got["counts"], ref["counts"])
    assert got["best_iter"] == ref["best_iter"] and got["best_count"] == ref["best_count"]
    assert np.array_equal(got["mask"], ref["mask"])
    assert np.allclose(got["Rt"], ref["Rt"], rtol=0, atol=1e-12)
    if n >= 500:
        assert np.abs(got["Rt"][:, :3] - R).max() < 5e-3 and np.abs(got["Rt"][:, 3] - t).max() < 5e-2   # estimator sanity
        inl = np.setdiff1d(np.arange(n), out)
        assert got["mask"][inl].mean() > 0.9 and got["mask"][out].mean() < 0.1


def test_solve_pnp_ransac_wrapper(ctx_small):
    from openvo_amd.ransac import solve_pnp_ransac
    X, uv, K4, R, t, out = _pnp_scene(800, 5, outlier_frac=0.2, noise=0.1)
    K = np.array([[K4[0], 0, K4[2]], [0, K4[1], K4[3]], [0, 0, 1.0]])
    Rg, tg, mask, info = solve_pnp_ransac(X, uv, K, iters=300, threshold=1.5, context=ctx_small)
    assert np.abs(Rg - R).max() < 3e-3 and np.abs(tg - t).max() < 3e-2 and info["best_count"] >= 0.7 * 800
    with pytest.raises(Exception):
        ctx_small.ransac_pnp(X[:3], uv[:3], K4, 10)


@pytest.mark.parametrize("w,h,ndisp,mind", [(333, 201, 48, 0), (621, 187, 64, -8), (257, 129, 16, 3)])
def test_odd_image_sizes_sgbm_and_orb(oracle, ctx_small, w, h, ndisp, mind):
    """Widths / heights that are multiples of nothing (KITTI's 1241x376 halves to 621x187): SGBM disparity
    and ORB keypoints + descriptors stay bit-exact."""
    c, L, R = _pair("C1", 4)
    L, R = np.ascontiguousarray(L[:h, :w]), np.ascontiguousarray(R[:h, :w])
    p = dict(minDisparity=mind, numDisparities=ndisp, blockSize=5, P1=200, P2=800, disp12MaxDiff=1, preFilterCap=63,
             uniquenessRatio=10, speckleWindowSize=100, speckleRange=2)
    for mode in (0, 1):
        ctx_small.set_sgbm(p, mode)
        got = ctx_small.sgbm_compute_host(L, R)
        ref = oracle.sgbm_compute(L, R, p, mode)
        assert np.array_equal(got, ref), "mode %d: %d pixels differ" % (mode, int((got != ref).sum()))
    g = ctx_small.orb_host(L, None, 300)
    r = oracle.orb_detect_and_compute(L, None, 300)
    assert len(g["xy"]) == len(r["xy"]) > 50
    assert np.array_equal(g["xy"].view(np.uint32), r["xy"].view(np.uint32)) and np.array_equal(g["desc"], r["desc"])
    assert np.array_equal(g["angle"].view(np.uint32), r["angle"].view(np.uint32)) and np.array_equal(g["octave"], r["octave"])


def test_point_clouds_generic_path_matches_reference_golden_g7():
    """The product's point_clouds through its public seams -- a user-supplied matcher (here: the scripted kNN tables of
    fixture G7), KeyPoint-like objects, host 3-D images -- against what the reference itself returned for the same
    inputs: the strict `<` of the ratio test at and around ratio * second, min_matches, queryIdx -> frame 1 /
    trainIdx -> frame 2, and the sampled points bit for bit (stereo_odometer.py:162-175)."""
    from openvo_amd import StereoCamera, StereoOdometer
    g = np.load(os.path.join(GOLD, "g7_point_clouds.npz"))
    names = sorted({k.split("__")[0] for k in g.files if "__" in k})
    c = Corridor("T0")
    cam = StereoCamera(c.K(), c.dist(), c.K(), c.dist(), c.rect_params(), c.sgbm_params(), (c.w, c.h), max_keypoints=300)

    class KP:
        def __init__(self, xy):
            self.pt = (float(xy[0]), float(xy[1]))

    class DM:
        def __init__(self, q, t, d):
            self.queryIdx, self.trainIdx, self.distance = int(q), int(t), float(d)

    class Matcher:
        def __init__(self, table):
            self.table = table

        def knnMatch(self, d1, d2, k=2):
            assert k == 2
            return self.table

    n_pts = 0
    for name in names:
        rows, train, second = g[name + "__rows"].reshape(-1, 2), g[name + "__train"], g[name + "__second"]
        table = [(DM(i, train[i], rows[i][0]), DM(i, second[i], rows[i][1])) for i in range(len(rows))]
        od = StereoOdometer(cam, match_threshold=float(g[name + "__match_threshold"]), min_matches=int(g[name + "__min_matches"]))
        od.matcher = Matcher(table)
        kp1 = [KP(p) for p in g[name + "__kp1"].reshape(-1, 2)]
        kp2 = [KP(p) for p in g[name + "__kp2"].reshape(-1, 2)]
        p1, p2 = od.point_clouds(kp1, kp2, "desc1", "desc2", g["im1"], g["im2"])
        if bool(g[name + "__none"]):
            assert p1 is None and p2 is None, name
            continue
        for got, want in ((p1, g[name + "__pts1"]), (p2, g[name + "__pts2"])):
            got = np.asarray(got)
            assert got.dtype == np.float32 and got.shape == want.shape, name
            assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), name
        n_pts += len(p1)
    assert n_pts > 80


def test_image_too_wide_for_the_fused_post_kernel_takes_the_separate_passes(oracle):
    """k_sgbm_post_rows keeps (RB + 2) image rows of disp1 / disp2 in LDS; an image wider than ~8500 pixels does not fit even
    one output row per block and falls back to the same steps as separate passes over HBM (k_sgbm_fin, k_lr_median3,
    k_ccl_rows).  Same bits either way."""
    from openvo_amd import _native
    w, h = 8704, 48
    rng = np.random.default_rng(5)
    base = rng.integers(0, 256, (h, w + 16), dtype=np.uint8)
    base = ((base.astype(np.int32) + np.roll(base, 1, 1) + np.roll(base, 1, 0)) // 3).astype(np.uint8)   # a little smoothing: real matches
    L, R = np.ascontiguousarray(base[:, :w]), np.ascontiguousarray(base[:, 5:5 + w])                         # L(x) = R(x - 5): disparity 5 px
    p = dict(minDisparity=0, numDisparities=16, blockSize=5, P1=100, P2=400, disp12MaxDiff=1, preFilterCap=31,
             uniquenessRatio=10, speckleWindowSize=60, speckleRange=2)
    ctx = _native.Context(0, w, 64, 16, 100)
    try:
        ctx.set_sgbm(p, 0)
        got = ctx.sgbm_compute_host(L, R)
        assert ctx.sgbm_last_schedule() in (_native.SCHED_DIAG, _native.SCHED_DIAG_RAGGED)
    finally:
        ctx.close()
    ref = oracle.sgbm_compute(L, R, p, 0)
    assert np.array_equal(got, ref), "%d pixels differ" % int((got != ref).sum())
    assert (ref == 80).mean() > 0.3            # 5 px x 16: the planted shift is what was found


def test_orb_selection_with_a_flood_of_score_ties(oracle):
    """retainBest keeps every candidate that ties with the n-th best (orb.cpp / KeyPointsFilter).  On a periodic texture (an
    8 x 8 tile repeated over the image) every corner exists 4800 times with one FAST score and one Harris response: the
    survivors of level 0 (3744 of them: quota 109 + its ties) no longer fit the fused selection kernel's LDS lists
    (ORB_SEL_CAP = 2048) and that level falls back to the global scratch arrays inside the same launch, while the other
    levels stay in LDS.  Same keypoints, responses, angles and descriptors as the oracle either way."""
    from openvo_amd import _native
    h, w = 480, 640
    tile = np.random.default_rng(0).choice([30, 220], size=(8, 8), p=[0.6, 0.4]).astype(np.uint8)
    img = np.tile(tile, (h // 8, w // 8))
    ctx = _native.Context(0, 704, 512, 64, 8000)           # capacity for the ties: far more than nfeatures keypoints come back
    try:
        got = ctx.orb_host(img, None, 500)
    finally:
        ctx.close()
    ref = oracle.orb_detect_and_compute(img, None, 500, cap=40000)
    assert np.bincount(ref["octave"], minlength=8)[0] > 2048, np.bincount(ref["octave"])    # level 0: the ties were kept (quota 109)
    assert np.array_equal(got["octave"], ref["octave"])
    assert np.array_equal(got["xy"].view(np.uint32), ref["xy"].view(np.uint32))
    assert np.array_equal(got["response"].view(np.uint32), ref["response"].view(np.uint32))
    assert np.array_equal(got["angle"].view(np.uint32), ref["angle"].view(np.uint32))
    assert np.array_equal(got["desc"], ref["desc"])


@pytest.mark.parametrize("w,h", [(63, 300), (300, 63), (64, 256), (1400, 66), (66, 1100), (67, 69), (62, 200)])
def test_orb_thin_images_keep_the_pixels_inside_the_border(oracle, w, h):
    """An image 63 .. 69 pixels wide or high still has 1 .. 7 columns / rows inside ORB's 31-pixel border at level 0
    (runByImageBorder keeps x in [31, w - 31)): the extraction must not give up on it (round 4: it returned nothing below
    70 pixels).  62 pixels: nothing is left, like OpenCV.  With and without a mask, few and many features."""
    from openvo_amd import _native
    from tests.big_pair import canvas
    img = canvas(w, h, seed=w * 7 + h)
    stripe = np.zeros((h, w), np.uint8)
    stripe[:, ::3] = 255
    stripe[::5] = 0
    ctx = _native.Context(0, 1408, 1104, 64, 3000)
    try:
        total = 0
        for nf in (7, 500, 3000):
            for m in (None, stripe):
                g = ctx.orb_host(img, m, nf)
                r = oracle.orb_detect_and_compute(img, m, nf, cap=20000)
                assert len(g["xy"]) == len(r["xy"]), (nf, m is not None)
                total += len(r["xy"])
                for k in ("xy", "response", "angle"):
                    assert np.array_equal(g[k].view(np.uint32), r[k].view(np.uint32)), (k, nf)
                assert np.array_equal(g["octave"], r["octave"]) and np.array_equal(g["desc"], r["desc"])
        assert (total > 0) == (min(w, h) > 62)
    finally:
        ctx.close()


def test_sgbm_every_sum_saturated_is_invalid_like_opencv(oracle):
    """P1 = 1000, P2 = 8000, blockSize 11, MODE_HH: where nothing matches, all 8 path costs of a pixel are large and every
    summed cost saturates at 32767.  OpenCV's winner search starts from minS = SHRT_MAX and compares with `<`: it finds nothing,
    bestDisp stays -1 and the pixel comes out as (minD - 1) * 16 -- INVALID -- whatever the uniqueness ratio (here 0).  Found by
    the parameter fuzz below (round 4: the kernels returned disparity 0 there)."""
    from openvo_amd import _native
    from tests.big_pair import pair
    p = dict(minDisparity=0, numDisparities=32, blockSize=11, P1=1000, P2=8000, disp12MaxDiff=1000, preFilterCap=1,
             uniquenessRatio=0, speckleWindowSize=0, speckleRange=0)
    L, R, _ = pair(333, 150, dmin=1, dmax=30, seed=3)
    R = np.ascontiguousarray(R[::-1])                   # rows swapped top to bottom: nothing matches anywhere
    ctx = _native.Context(0, 512, 200, 32, 64)
    try:
        hit = 0
        for mode in (0, 1):
            ctx.set_sgbm(p, mode)
            got = ctx.sgbm_compute_host(L, R)
            ref = oracle.sgbm_compute(L, R, p, mode)
            assert np.array_equal(got, ref), "mode %d: %d pixels differ" % (mode, int((got != ref).sum()))
            hit += int((ref[:, 40:] == -16).sum())
        assert hit > 0                                   # the case is really exercised: invalid pixels right of the left band
    finally:
        ctx.close()


@pytest.mark.parametrize("seed", [11, 12])
def test_sgbm_parameter_fuzz_against_the_oracle(oracle, seed):
    """150 random draws per seed over everything StereoSGBM_create takes -- image size, numDisparities 16 .. 256, minDisparity
    -32 .. 40, blockSize 1 .. 11, P1 / P2 down to 0 / equal and up to the limit, preFilterCap 0 .. 127, uniquenessRatio 0 ..
    100, speckle window 0 .. larger than the image, speckle range 0 .. 100, disp12MaxDiff -1 .. 1000, both modes; textured,
    unmatched and textureless pairs: bit-exact, or refused with the documented message (a path cost beyond int16)."""
    from openvo_amd import _native
    from tests.big_pair import pair
    rng = np.random.default_rng(seed)
    ctxs, refused = {}, 0
    try:
        for it in range(150):
            w = int(rng.choice([180, 257, 333, 400, 512])); h = int(rng.choice([64, 97, 150, 200]))
            D = int(rng.choice([16, 32, 48, 64, 96, 128, 160, 256]))
            if w - D < 20:
                D = 16
            mind = int(rng.choice([-32, -1, 0, 0, 7, 40]))
            P1 = int(rng.choice([0, 1, 8, 200, 1000]))
            p = dict(minDisparity=mind, numDisparities=D, blockSize=int(rng.choice([1, 3, 5, 5, 7, 9, 11])), P1=P1,
                     P2=min(8000, P1 + int(rng.choice([0, 1, 24, 600, 3000, 7000]))), disp12MaxDiff=int(rng.choice([-1, 0, 1, 5, 1000])),
                     preFilterCap=int(rng.choice([0, 1, 15, 31, 63, 100, 127])), uniquenessRatio=int(rng.choice([0, 5, 15, 50, 99, 100])),
                     speckleWindowSize=int(rng.choice([0, 10, 100, 100000])), speckleRange=int(rng.choice([0, 1, 2, 10, 100])))
            mode = int(rng.integers(0, 2))
            L, R, _ = pair(w, h, dmin=max(1, mind + 1), dmax=max(2, min(D + mind - 2, w // 3)), seed=int(rng.integers(1, 1000)))
            if rng.random() < 0.2:
                R = np.ascontiguousarray(L[:, ::-1])
            if rng.random() < 0.15:
                L = np.full_like(L, 77); R = np.full_like(R, 77)
            if D not in ctxs:
                ctxs[D] = _native.Context(0, 512, 200, D, 64)
            try:
                ctxs[D].set_sgbm(p, mode)
            except _native.VoError as e:
                assert "beyond the int16" in str(e), (p, str(e))
                refused += 1
                continue
            got = ctxs[D].sgbm_compute_host(L, R)
            ref = oracle.sgbm_compute(L, R, p, mode)
            assert np.array_equal(got, ref), "draw %d %s mode %d: %d pixels differ" % (it, p, mode, int((got != ref).sum()))
        assert refused < 30
    finally:
        for c in ctxs.values():
            c.close()


@pytest.mark.parametrize("seed", [31, 32])
def test_small_operator_fuzz_against_the_oracle(oracle, seed):
    """Hamming kNN-2 (1 .. 2000 descriptors a side, random / four distinct descriptors = ties everywhere / all-zero against
    all-one / near-copies) + the ratio filter at 0 .. 1.5, bilinear lookups in float images with inf / -inf / nan taps, at
    integer positions and at the last pixel, reprojectImageTo3D with zero and negative disparities, BGR -> gray, the rigid
    clique at thresholds 0 .. 1 with repeated and colinear points: bit for bit; Umeyama (1e-9) and Rodrigues (1e-12, rotations
    by pi and not-quite-rotations included); degenerate fits raise on both sides."""
    from openvo_amd import _native, calib
    rng = np.random.default_rng(seed)
    ctx = _native.Context(0, 640, 480, 64, 2000)
    try:
        for it in range(40):
            nq = int(rng.choice([1, 2, 3, 63, 64, 65, 200, 511, 512, 513, 2000])); nt = int(rng.choice([1, 2, 3, 63, 64, 65, 129, 1000, 2000]))
            kind = int(rng.integers(0, 4))
            if kind == 0:
                q = rng.integers(0, 256, (nq, 32), dtype=np.uint8); t = rng.integers(0, 256, (nt, 32), dtype=np.uint8)
            elif kind == 1:
                base = rng.integers(0, 256, (4, 32), dtype=np.uint8)
                q = base[rng.integers(0, 4, nq)]; t = base[rng.integers(0, 4, nt)]
            elif kind == 2:
                q = np.zeros((nq, 32), np.uint8); t = np.full((nt, 32), 255, np.uint8)
            else:
                t = rng.integers(0, 256, (nt, 32), dtype=np.uint8); q = t[rng.integers(0, nt, nq)].copy(); q[:, 0] ^= rng.integers(0, 4, nq).astype(np.uint8)
            gi, gd = ctx.bf_knn2(q, t)
            ri, rd = oracle.bf_knn2_hamming(q, t)
            assert np.array_equal(gi, ri) and np.array_equal(gd, rd), ("knn2", nq, nt, kind)
            if nt >= 2:
                for ratio in (0.0, 0.5, 0.8, 1.0, 1.5):
                    a, b = ctx.ratio_filter(gi, gd, ratio), oracle.ratio_filter(ri, rd, ratio)
                    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]), ("ratio", nq, nt, kind, ratio)
        for it in range(30):
            h, w = int(rng.integers(2, 40)), int(rng.integers(2, 50))
            img = (rng.normal(size=(h, w, 3)) * 50).astype(np.float32)
            img[rng.random((h, w)) < 0.2] = np.inf
            img[rng.random((h, w)) < 0.05] = -np.inf
            img[rng.random((h, w)) < 0.03] = np.nan
            n = int(rng.integers(1, 300))
            xy = np.stack([rng.uniform(0, w - 1, n), rng.uniform(0, h - 1, n)], 1).astype(np.float32)
            xy[: n // 4] = np.floor(xy[: n // 4])
            xy[0] = [w - 1, h - 1]
            g, gs = ctx.bilinear_at(img, xy)
            r, rs = oracle.bilinear_at(img, xy)
            assert np.array_equal(gs, rs) and np.array_equal(g.view(np.uint32)[rs != 2], r.view(np.uint32)[rs != 2]), ("bilinear", h, w, n)
        for it in range(15):
            h, w = int(rng.integers(1, 60)), int(rng.integers(1, 80))
            disp = (rng.integers(-16, 2000, (h, w)) / 16.0).astype(np.float32)
            disp[rng.random((h, w)) < 0.1] = 0
            Q = np.array([[1, 0, 0, -rng.uniform(10, 40)], [0, 1, 0, -rng.uniform(5, 30)], [0, 0, 0, rng.uniform(100, 900)],
                          [0, 0, 1 / rng.uniform(0.05, 0.6), rng.uniform(-1, 1)]])
            assert np.array_equal(ctx.reproject_to_3d(disp, Q).view(np.uint32), oracle.reproject_to_3d(disp, Q).view(np.uint32)), ("reproject", h, w)
            bgr = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
            assert np.array_equal(ctx.cvt_bgr2gray(bgr), oracle.bgr2gray(bgr))
        for it in range(40):
            n = int(rng.choice([3, 4, 10, 11, 63, 64, 65, 200, 513, 700]))
            P = (rng.normal(size=(n, 3)) * 5 + [0, 0, 20]).astype(np.float32)
            Rm = calib.rodrigues_vec_to_mat(rng.uniform(-0.2, 0.2, 3))
            C = (P @ Rm.T + rng.uniform(-1, 1, 3)).astype(np.float32)
            out = rng.random(n) < rng.choice([0.0, 0.2, 0.6])
            C[out] += (rng.normal(size=(int(out.sum()), 3)) * 3).astype(np.float32)
            if it % 7 == 0:
                P[:] = P[0]; C[:] = C[0]
            if it % 11 == 0:
                P[:, 1:] = 0; C[:, 1:] = 0
            thr = float(rng.choice([0.0, 0.01, 0.1, 1.0]))
            assert np.array_equal(np.asarray(ctx.rigid_clique(P, C, thr)) != 0, np.asarray(oracle.rigid_clique(P, C, thr)) != 0), ("clique", n, thr, it)
            for force in (True, False):
                res = []
                for f in (ctx.umeyama, oracle.umeyama):
                    try:
                        res.append(f(P, C, force))
                    except Exception:
                        res.append(None)
                assert (res[0] is None) == (res[1] is None), ("umeyama raises on one side only", n, it)
                if res[0] is not None:
                    assert np.allclose(res[0][0], res[1][0], rtol=0, atol=1e-9, equal_nan=True), ("umeyama", n, it, force)
        for it in range(120):
            Rm = calib.rodrigues_vec_to_mat(rng.normal(size=3) * rng.choice([1e-12, 1e-6, 0.1, 1.0, 3.1, 3.14159]))
            if it % 9 == 0:
                Rm = np.diag(rng.permutation([1.0, -1.0, -1.0]))
            if it % 13 == 0:
                Rm = Rm + rng.normal(size=(3, 3)) * 1e-3
            g, r = _native.Context.rodrigues(Rm), oracle.rodrigues(Rm)
            assert np.allclose(np.asarray(g).ravel(), np.asarray(r).ravel(), rtol=0, atol=1e-12, equal_nan=True), ("rodrigues", it)
    finally:
        ctx.close()


@pytest.mark.parametrize("seed", [41, 42])
def test_orb_fuzz_against_the_oracle(oracle, seed):
    """60 draws per seed: image sizes 63 .. 700 a side, seven kinds of content (blocky noise, white noise, checkerboards and
    tiled patterns = floods of equal scores, constant, ramps, binary), nfeatures 1 .. 4000 (both selection paths: <= 2000 fused,
    above in three launches), random / empty masks with values 1 or 255: keypoints, responses, angles, octaves and descriptors
    bit for bit -- or, when the ties push the count past the context's capacity, the documented refusal."""
    from openvo_amd import _native
    from tests.big_pair import canvas
    rng = np.random.default_rng(seed)
    ctx = _native.Context(0, 704, 704, 64, 4000)
    try:
        compared = 0
        for it in range(60):
            w, h = int(rng.integers(63, 700)), int(rng.integers(63, 700))
            kind = int(rng.integers(0, 7))
            yy, xx = np.mgrid[0:h, 0:w]
            if kind == 0:
                img = canvas(w, h, seed=int(rng.integers(1, 10000)))
            elif kind == 1:
                img = rng.integers(0, 256, (h, w), dtype=np.uint8)
            elif kind == 2:
                img = (((xx // int(rng.integers(2, 9))) + (yy // int(rng.integers(2, 9)))) % 2 * 200 + 20).astype(np.uint8)
            elif kind == 3:
                img = np.full((h, w), int(rng.integers(0, 256)), np.uint8)
            elif kind == 4:
                img = ((xx * 3 + yy * 5) % 256).astype(np.uint8)
            elif kind == 5:
                img = np.where(canvas(w, h, seed=int(rng.integers(1, 10000))) > 128, 255, 0).astype(np.uint8)
            else:
                t = rng.integers(0, 256, (int(rng.integers(3, 12)), int(rng.integers(3, 12))), dtype=np.uint8)
                img = np.tile(t, (h // t.shape[0] + 1, w // t.shape[1] + 1))[:h, :w].copy()
            nf = int(rng.choice([1, 2, 10, 100, 500, 1999, 2000, 2001, 4000]))
            m, r = None, rng.random()
            if r < 0.3:
                m = (rng.random((h, w)) > rng.uniform(0.05, 0.95)).astype(np.uint8) * int(rng.choice([1, 255]))
            elif r < 0.4:
                m = np.zeros((h, w), np.uint8)
            ref = oracle.orb_detect_and_compute(img, m, nf, cap=200000)
            try:
                got = ctx.orb_host(img, m, nf)
            except _native.VoError as e:
                assert "exceed capacity" in str(e) and len(ref["xy"]) > ctx.kp_cap, (it, str(e))
                continue
            compared += 1
            assert len(got["xy"]) == len(ref["xy"]), (it, w, h, kind, nf)
            for k in ("xy", "response", "angle"):
                assert np.array_equal(got[k].view(np.uint32), ref[k].view(np.uint32)), (k, it, w, h, kind, nf)
            assert np.array_equal(got["octave"], ref["octave"]) and np.array_equal(got["desc"], ref["desc"]), (it, w, h, kind, nf)
        assert compared >= 50
    finally:
        ctx.close()


@pytest.mark.parametrize("seed", [61])
def test_ransac_fuzz_against_the_cpu_restatement(oracle, seed):
    """Essential-matrix RANSAC (five-point and eight-point) and solvePnP RANSAC on random scenes: 5 .. 3000 correspondences,
    planar scenes, pure rotations, no motion at all, one correspondence repeated, 0 .. 70 % outliers, 1 .. 300 hypotheses,
    thresholds 0.1 .. 5 px: every hypothesis' inlier count, the winner, the mask bit for bit, E / [R|t] to 1e-11 / 1e-10; too
    few points raise on both sides.  (No openVO counterpart: parity against the build's own restatement)"""
    from openvo_amd import _native, calib
    rng = np.random.default_rng(seed)
    K4 = np.array([520.0, 515.0, 320.0, 240.0])

    def scene(n, planar, pure_rot, outl, noise):
        X = np.stack([rng.uniform(-4, 4, n), rng.uniform(-3, 3, n), rng.uniform(4, 20, n)], 1)
        if planar:
            X[:, 2] = 8 + 0.1 * X[:, 0]
        R = calib.rodrigues_vec_to_mat(rng.uniform(-0.1, 0.1, 3))
        t = np.zeros(3) if pure_rot else rng.uniform(-0.5, 0.5, 3)
        proj = lambda P: np.stack([K4[0] * P[:, 0] / P[:, 2] + K4[2], K4[1] * P[:, 1] / P[:, 2] + K4[3]], 1)
        p1, p2 = proj(X), proj(X @ R.T + t) + rng.normal(size=(n, 2)) * noise
        o = rng.random(n) < outl
        p2[o] = rng.uniform(0, 640, (int(o.sum()), 2))
        return X.astype(np.float32), p1.astype(np.float32), p2.astype(np.float32)

    def both(f, g):
        out = []
        for fn in (f, g):
            try:
                out.append(fn())
            except Exception:
                out.append(None)
        return out

    ctx = _native.Context(0, 640, 480, 64, 4000)
    try:
        for it in range(40):
            n = int(rng.choice([5, 6, 7, 8, 9, 20, 64, 65, 500, 3000]))
            X, p1, p2 = scene(n, rng.random() < 0.2, rng.random() < 0.2, float(rng.choice([0, 0.2, 0.7])), float(rng.choice([0, 0.3, 2.0])))
            if it % 10 == 0:
                p2 = p1.copy()
            if it % 13 == 0:
                p1[:] = p1[0]; p2[:] = p2[0]
            iters, thr, sd = int(rng.choice([1, 63, 64, 65, 300])), float(rng.choice([0.1, 1.0, 5.0])), int(rng.integers(0, 2**31))
            for solver in (5, 8):
                g, r = both(lambda: ctx.ransac_essential(p1, p2, K4, iters, thr, sd, want_counts=True, solver=solver),
                            lambda: oracle.ransac_essential(p1, p2, K4, iters, thr, sd, solver=solver))
                assert (g is None) == (r is None), ("essential raises on one side only", solver, n)
                if g is not None:
                    assert np.array_equal(g["counts"], r["counts"]) and g["best_iter"] == r["best_iter"] and np.array_equal(g["mask"], r["mask"]), (solver, n, it)
                    assert np.allclose(g["E"], r["E"], rtol=0, atol=1e-11, equal_nan=True), (solver, n, it)
            g, r = both(lambda: ctx.ransac_pnp(X, p2, K4, iters, thr, sd, want_counts=True), lambda: oracle.ransac_pnp(X, p2, K4, iters, thr, sd))
            assert (g is None) == (r is None), ("pnp raises on one side only", n)
            if g is not None:
                assert np.array_equal(g["counts"], r["counts"]) and g["best_iter"] == r["best_iter"] and np.array_equal(g["mask"], r["mask"]), ("pnp", n, it)
                assert np.allclose(g["Rt"], r["Rt"], rtol=0, atol=1e-10, equal_nan=True), ("pnp", n, it)
    finally:
        ctx.close()
