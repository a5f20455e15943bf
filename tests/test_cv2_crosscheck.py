"""Opportunistic cross-check against a REAL OpenCV (SURVEY section 4 tier 3).

cv2 is the reference's only arithmetic dependency and it is neither vendored nor installed in the build
container, so everywhere else parity is judged against oracle/ (a CPU restatement) -- "parity unpinned".  This
file pins it the moment a cv2 is importable: the same inputs go through the cv2 calls the reference makes
(/root/reference/src/openVO/stereo_camera.py:17-27,43-55; stereo_odometer.py:117,163,190,204,212 -- issued by this
harness, the reference's files are not used) and through (a) the oracle, on CPU, and (b) the HIP path behind the
C ABI, on the GPU.  Without cv2 every test here is skipped; that is expected."""
import numpy as np
import pytest

cv2 = pytest.importorskip("cv2")

from openvo_amd import calib                                    # noqa: E402
from openvo_amd.synth import Corridor                           # noqa: E402

SGBM = dict(minDisparity=0, numDisparities=64, blockSize=5, P1=200, P2=800, disp12MaxDiff=1, preFilterCap=63,
            uniquenessRatio=10, speckleWindowSize=100, speckleRange=2)
RIG = dict(K1=np.array([[498.0, 0, 325.0], [0, 502.0, 236.0], [0, 0, 1]]), K2=np.array([[507.0, 0, 314.0], [0, 511.0, 244.0], [0, 0, 1]]),
           d1=np.array([-0.11, 0.03, 0.0007, -0.0004, 0.002]), d2=np.array([-0.09, 0.015, -0.0003, 0.0006, -0.001]),
           R=calib.rodrigues_vec_to_mat([0.012, -0.021, 0.008]), T=np.array([-0.2, 0.004, -0.003]))


def _cv_sgbm(p, mode=0):
    return cv2.StereoSGBM_create(p["minDisparity"], p["numDisparities"], p["blockSize"], p["P1"], p["P2"], p["disp12MaxDiff"],
                                 p["preFilterCap"], p["uniquenessRatio"], p["speckleWindowSize"], p["speckleRange"], mode=mode)


def _kp_key(xy, octave):
    return sorted(zip(octave.tolist(), np.asarray(xy)[:, 1].tolist(), np.asarray(xy)[:, 0].tolist()))


def _cv_orb(img, mask, n):
    kps, desc = cv2.ORB_create(nfeatures=n).detectAndCompute(img, mask)
    xy = np.array([k.pt for k in kps], np.float32).reshape(-1, 2)
    octv = np.array([k.octave for k in kps], np.int32)
    order = sorted(range(len(kps)), key=lambda i: (octv[i], xy[i, 1], xy[i, 0]))     # compare as sets
    return xy[order], octv[order], np.array([kps[i].angle for i in order], np.float32), \
        np.array([kps[i].response for i in order], np.float32), desc[order]


@pytest.fixture(scope="module")
def pair():
    c = Corridor("C1")
    return c, c.pair(2)


# ---- (a) the oracle against cv2, on CPU ------------------------------------------------------------------------
def test_oracle_stereo_rectify_and_maps_equal_cv2(oracle):
    """cv2.stereoRectify / initUndistortRectifyMap against BOTH restatements: the checker (oracle/src/calib.c) and the
    product's host code (openvo_amd/calib.py) -- tests/test_calib_known_answers.py holds the two against each other."""
    ref = cv2.stereoRectify(RIG["K1"], RIG["d1"], RIG["K2"], RIG["d2"], (640, 480), RIG["R"], RIG["T"])
    for impl, who in ((oracle, "oracle/src/calib.c"), (calib, "openvo_amd/calib.py")):
        got = impl.stereo_rectify(RIG["K1"], RIG["d1"], RIG["K2"], RIG["d2"], (640, 480), RIG["R"], RIG["T"])
        for g, r, name in zip(got[:5], ref[:5], ("R1", "R2", "P1", "P2", "Q")):
            assert np.allclose(g, r, rtol=0, atol=1e-9), (who, name)
        assert tuple(got[5]) == tuple(ref[5]) and tuple(got[6]) == tuple(ref[6]), who
        for K, d, R, P in ((RIG["K1"], RIG["d1"], ref[0], ref[2]), (RIG["K2"], RIG["d2"], ref[1], ref[3])):
            m1, m2 = impl.init_undistort_rectify_map(K, d, R, P, (640, 480))
            c1, c2 = cv2.initUndistortRectifyMap(K, d, R, P, (640, 480), cv2.CV_16SC2)
            # float rounding of u*32 may flip a fixed-point LSB on isolated pixels; nothing larger
            assert (np.abs(m1.astype(int) - c1.astype(int)).max() <= 1) and ((m1 != c1).any(-1).mean() < 1e-3), who
            assert (m2 != c2).mean() < 1e-3, who


def test_oracle_image_stages_equal_cv2(oracle, pair):
    c, (L, R) = pair
    bgr = np.stack([L, np.roll(L, 3, 0), 255 - L], -1)
    assert np.array_equal(oracle.bgr2gray(bgr), cv2.cvtColor(bgr, cv2.COLOR_BGR2GRAY))
    R1, R2, P1, P2, Q, _, _ = cv2.stereoRectify(c.K(), RIG["d1"], c.K(), RIG["d2"], (c.w, c.h), RIG["R"], np.array([-c.B, 0.001, 0.0]))
    m1, m2 = cv2.initUndistortRectifyMap(c.K(), RIG["d1"], R1, P1, (c.w, c.h), cv2.CV_16SC2)
    assert np.array_equal(oracle.remap_bilinear(L, m1, m2), cv2.remap(L, m1, m2, cv2.INTER_LINEAR))
    for mode in (0, 1):
        ref = _cv_sgbm(SGBM, mode).compute(L, R)
        got = oracle.sgbm_compute(L, R, SGBM, mode)
        assert np.array_equal(got, ref), "mode %d: %d pixels differ" % (mode, int((got != ref).sum()))
    d = ref.astype(np.float32) / 16
    with np.errstate(all="ignore"):
        assert np.array_equal(oracle.reproject_to_3d(d, c.Q()).view(np.uint32), cv2.reprojectImageTo3D(d, c.Q()).view(np.uint32))


def test_oracle_features_and_pose_equal_cv2(oracle, pair):
    c, (L, R) = pair
    d = _cv_sgbm(SGBM).compute(L, R).astype(np.float32) / 16
    mask = ((d >= 4) * (d <= 100)).astype(np.uint8) * 255
    for m in (mask, None):
        xy, octv, ang, resp, desc = _cv_orb(L, m, 500)
        o = oracle.orb_detect_and_compute(L, m, 500)
        assert _kp_key(o["xy"], o["octave"]) == _kp_key(xy, octv)
        assert np.array_equal(o["desc"], desc) and np.allclose(o["angle"], ang, atol=1e-3) and np.allclose(o["response"], resp, rtol=1e-5)
    a = oracle.orb_detect_and_compute(L, None, 500)["desc"]
    b = oracle.orb_detect_and_compute(R, None, 500)["desc"]
    m = cv2.BFMatcher.create(cv2.NORM_HAMMING).knnMatch(a, b, k=2)
    idx, dist = oracle.bf_knn2_hamming(a, b)
    assert [(x.trainIdx, y.trainIdx) for x, y in m] == [tuple(r) for r in idx.tolist()]
    assert [(x.distance, y.distance) for x, y in m] == [tuple(map(float, r)) for r in dist.tolist()]
    rng = np.random.default_rng(0)
    src = rng.normal(size=(60, 3)).astype(np.float32) * 3
    Rm = calib.rodrigues_vec_to_mat([0.02, -0.05, 0.01])
    dst = (src @ Rm.T + [0.1, -0.05, 0.3] + rng.normal(scale=0.01, size=src.shape)).astype(np.float32)
    T, s = cv2.estimateAffine3D(src, dst, force_rotation=True)
    To, so = oracle.umeyama(src, dst, True)
    assert np.allclose(To, T, rtol=0, atol=1e-9) and abs(so - s) < 1e-9
    assert np.allclose(oracle.rodrigues(T[:, :3]).ravel(), cv2.Rodrigues(T[:, :3])[0].ravel(), atol=1e-12)


# ---- (b) the HIP path against cv2, on the GPU ------------------------------------------------------------------
@pytest.mark.gpu
def test_hip_path_equals_cv2(pair):
    from openvo_amd import StereoCamera, StereoOdometer
    c, (L, R) = pair
    cam = StereoCamera(c.K(), c.dist(), c.K(), c.dist(), c.rect_params(), SGBM, (c.w, c.h), max_keypoints=500)
    for mode in (0, 1):
        cam._ctx.set_sgbm(SGBM, mode)
        assert np.array_equal(cam._ctx.sgbm_compute_host(L, R), _cv_sgbm(SGBM, mode).compute(L, R))
    cam._ctx.set_sgbm(SGBM, 0)
    x3, disp, left = cam.compute_3d(L, R, preprocessed=True)
    d = _cv_sgbm(SGBM).compute(L, R).astype(np.float32) / 16
    vr = cam.valid_region_left
    with np.errstate(all="ignore"):
        ref3 = cv2.reprojectImageTo3D(d, cam.Q)[vr[1]:vr[3], vr[0]:vr[2]]
    assert np.array_equal(np.asarray(x3).view(np.uint32), ref3.view(np.uint32))
    odo = StereoOdometer(cam, preprocessed_frames=True)
    kps, desc = odo.orb.detectAndCompute(left, odo.feature_mask(disp))
    dm = d[vr[1]:vr[3], vr[0]:vr[2]]
    xy, octv, ang, resp, cdesc = _cv_orb(np.ascontiguousarray(L[vr[1]:vr[3], vr[0]:vr[2]]), ((dm >= 4) * (dm <= 100)).astype(np.uint8) * 255, 500)
    assert _kp_key(kps.xy, kps.octave) == _kp_key(xy, octv) and np.array_equal(desc, cdesc)
    m = cv2.BFMatcher.create(cv2.NORM_HAMMING).knnMatch(cdesc, cdesc[::-1].copy(), k=2)
    mine = odo.matcher.knnMatch(desc, desc[::-1].copy(), k=2)
    assert [(a.trainIdx, b.trainIdx, a.distance, b.distance) for a, b in m] == [(a.trainIdx, b.trainIdx, a.distance, b.distance) for a, b in mine]
    # the whole chain: ten frames, chained pose within 1e-4 m of the cv2 path (BASELINE target)
    import bench
    frames = c.pairs(0, 6)
    with bench._cv2_seams(cv2):
        ref_poses = bench._cv2_chunk(cv2, c, cam, SGBM, frames, 1, 5)
    odo = StereoOdometer(cam, **bench.ODO_KW)
    for Lk, Rk in frames:
        assert odo.update(Lk, Rk)
    assert np.linalg.norm(odo.current_pose()[:3, 3] - ref_poses[-1][:3, 3]) <= 1e-4
