"""End-to-end GPU parity: StereoCamera / StereoOdometer (HIP behind the C ABI) against the CPU
oracle pipeline on the same synthetic sequences, plus size-independent properties at the full
BASELINE sizes."""
import numpy as np
import pytest

from openvo_amd import StereoCamera, StereoOdometer, _native
from openvo_amd.synth import Corridor

pytestmark = pytest.mark.gpu


def _rig(name, **kw):
    c = Corridor(name)
    cam = StereoCamera(c.K(), c.dist(), c.K(), c.dist(), c.rect_params(), c.sgbm_params(), (c.w, c.h), **kw)
    return c, cam


def _ref(c, cam, **kw):
    from oracle.odometer import RefStereoCamera, RefStereoOdometer
    rcam = RefStereoCamera(cam.Q, cam.valid_region_left, c.sgbm_params())
    return rcam, RefStereoOdometer(rcam, **kw)


@pytest.mark.parametrize("kw", [dict(), dict(rigidity_threshold=0.1, outlier_threshold=0.02)])
def test_c1_sequence_matches_oracle(kw):
    """BASELINE config 1: 10 synthetic 640x480 pairs.  Every frame: accept flag, skip_cause,
    disparity, keypoint set, descriptors identical; chained pose within 1e-9 (float64 sums are
    reduced in a different order on the GPU)."""
    c, cam = _rig("C1", max_keypoints=500)
    odo = StereoOdometer(cam, preprocessed_frames=True, **kw)
    rcam, rodo = _ref(c, cam, preprocessed_frames=True, **kw)
    vr = cam.valid_region_left
    for k in range(10):
        L, R = c.pair(k)
        a, b = odo.update(L, R), rodo.update(L, R)
        assert a == b and odo.skip_cause == rodo.skip_cause and odo.skipped_frames == rodo.skipped_frames, k
        if a:
            d16 = np.rint(np.asarray(odo.current_disparity) * 16).astype(np.int16)
            assert np.array_equal(d16, rcam.last_disp16[vr[1]:vr[3], vr[0]:vr[2]])
            assert np.array_equal(odo.current_kps.xy.view(np.uint32), rodo.cur["kps"]["xy"].view(np.uint32))
            assert np.array_equal(odo.current_kps.octave, rodo.cur["kps"]["octave"])
            assert np.array_equal(odo.current_desc, rodo.cur["desc"])
        assert np.allclose(odo.c_T_w, rodo.c_T_w, rtol=0, atol=1e-9), k
    ate = np.sqrt(np.mean([np.sum((odo.current_pose()[:3, 3] - rodo.current_pose()[:3, 3]) ** 2)]))
    assert ate <= 1e-4


def test_compute_3d_outputs_and_cv2_seams(oracle):
    """compute_3d's three lazily materialised return values and each cv2-object seam."""
    c, cam = _rig("T0", max_keypoints=500)
    L, R = c.pair(2)
    img_3d, disp, left = cam.compute_3d(L, R, preprocessed=True)
    vr = cam.valid_region_left
    ref16 = oracle.sgbm_compute(L, R, c.sgbm_params(), 0)
    refd = (ref16.astype(np.float32) / 16)
    assert disp.shape == refd[vr[1]:vr[3], vr[0]:vr[2]].shape and disp.dtype == np.float32
    assert np.array_equal(np.asarray(disp), refd[vr[1]:vr[3], vr[0]:vr[2]])
    assert np.array_equal(np.asarray(left), L[vr[1]:vr[3], vr[0]:vr[2]])
    with np.errstate(all="ignore"):
        ref3 = oracle.reproject_to_3d(refd, cam.Q)[vr[1]:vr[3], vr[0]:vr[2]]
    got3 = np.asarray(img_3d)
    assert got3.shape == ref3.shape and got3.dtype == np.float32
    assert np.array_equal(got3.view(np.uint32), ref3.view(np.uint32))       # incl. inf / nan patterns
    # stereoSGBM.compute seam
    assert np.array_equal(cam.stereoSGBM.compute(L, R), ref16)
    # orb.detectAndCompute / matcher.knnMatch seams on plain numpy arrays
    odo = StereoOdometer(cam, nfeatures=300)
    mask = odo.feature_mask(np.asarray(disp))
    kps, desc = odo.orb.detectAndCompute(np.asarray(left), mask)
    ref = oracle.orb_detect_and_compute(np.asarray(left), mask, 300)
    assert np.array_equal(desc, ref["desc"]) and np.array_equal(kps.xy, ref["xy"])
    assert isinstance(kps[0].pt, tuple) and isinstance(kps[0].pt[0], float)
    # the fused device path gives the same keypoints as the host-array path
    kps2, desc2 = odo.orb.detectAndCompute(left, odo.feature_mask(disp))
    assert np.array_equal(desc2, desc) and np.array_equal(kps2.xy, kps.xy)
    m = odo.matcher.knnMatch(desc, desc[::-1].copy(), k=2)
    ri, rd = oracle.bf_knn2_hamming(desc, desc[::-1].copy())
    assert [x[0].trainIdx for x in m] == ri[:, 0].tolist() and [x[1].distance for x in m] == rd[:, 1].astype(float).tolist()
    # bilinear_interpolate_pixels seam: device image and host array agree with the oracle
    xy = kps.xy[:20]
    refp, _ = oracle.bilinear_at(ref3, xy)
    for i in range(len(xy)):
        p = odo.bilinear_interpolate_pixels(img_3d, float(xy[i, 0]), float(xy[i, 1]))
        q = odo.bilinear_interpolate_pixels(got3, float(xy[i, 0]), float(xy[i, 1]))
        assert np.array_equal(p, refp[i]) and np.array_equal(q, refp[i])


def test_unrectified_colour_input(oracle):
    """cvtColor + remap front-end (preprocessed=False, BGR input) against the oracle."""
    from openvo_amd import calib
    c = Corridor("T0")
    dist = np.array([-0.08, 0.01, 0.0005, -0.0003, 0.0])
    rect = {"R": calib.rodrigues_vec_to_mat([0.002, 0.004, -0.003]), "T": np.array([-c.B, 0.001, 0.0])}
    cam = StereoCamera(c.K(), dist, c.K(), dist, rect, c.sgbm_params(), (c.w, c.h), max_keypoints=300)
    L, R = c.pair(1)
    Lc = np.stack([L, np.roll(L, 1, 0), 255 - L], -1)
    Rc = np.stack([R, np.roll(R, 1, 0), 255 - R], -1)
    _, disp, left = cam.compute_3d(Lc, Rc)
    gl = oracle.remap_bilinear(oracle.bgr2gray(Lc), cam.map_left_1, cam.map_left_2)
    gr = oracle.remap_bilinear(oracle.bgr2gray(Rc), cam.map_right_1, cam.map_right_2)
    vr = cam.valid_region_left
    assert np.array_equal(np.asarray(left), gl[vr[1]:vr[3], vr[0]:vr[2]])
    assert np.array_equal(cam.undistort_rectify_right(oracle.bgr2gray(Rc)), gr)
    ref16 = oracle.sgbm_compute(gl, gr, c.sgbm_params(), 0)
    assert np.array_equal(np.rint(np.asarray(disp) * 16).astype(np.int16), ref16[vr[1]:vr[3], vr[0]:vr[2]])


def test_raw_colour_pairs_through_run_equal_plain_updates():
    """The reference's default input: unrectified BGR pairs (preprocessed_frames=False).  Handed to run() they are copied into
    pinned staging by the library's thread, read from there by the ingest kernel (3 bytes per pixel), converted and remapped on a
    look-ahead engine -- poses bit-identical to plain update(left, right) calls on the same arrays."""
    from openvo_amd import calib
    c = Corridor("C1")
    dist = np.array([-0.05, 0.01, 0.0003, -0.0002, 0.0])
    rect = {"R": calib.rodrigues_vec_to_mat([0.001, 0.002, -0.001]), "T": np.array([-c.B, 0.0, 0.0])}
    cam = StereoCamera(c.K(), dist, c.K(), dist, rect, c.sgbm_params(), (c.w, c.h), max_keypoints=500)
    frames = []
    for L, R in c.pairs(0, 8):
        frames.append((np.ascontiguousarray(np.stack([L, L, L], -1)), np.ascontiguousarray(np.stack([R, R, R], -1))))
    kw = dict(rigidity_threshold=0.1, outlier_threshold=0.02)
    plain = StereoOdometer(cam, **kw)
    want = [(plain.update(L, R), plain.c_T_w.copy()) for L, R in frames]
    odo = StereoOdometer(cam, **kw)
    got = [(ok, odo.c_T_w.copy()) for ok in odo.run(iter(frames), depth=4)]
    assert len(got) == len(want) and sum(a for a, _ in got) >= 6
    for (a, Ta), (b, Tb) in zip(got, want):
        assert a == b and np.array_equal(Ta, Tb)


def test_frame_eviction_keeps_results():
    """Holding more frames than device slots moves the oldest to host memory transparently."""
    c, cam = _rig("T0", max_keypoints=300)
    outs = []
    for k in range(6):
        L, R = c.pair(k)
        outs.append(cam.compute_3d(L, R, preprocessed=True))
    first = np.asarray(outs[0][1])
    fresh = np.asarray(cam.compute_3d(*c.pair(0), preprocessed=True)[1])
    assert np.array_equal(first, fresh)


@pytest.mark.parametrize("name,mode,nfeat", [("C2", 0, 500), ("C4", 1, 500)])
def test_full_size_properties(name, mode, nfeat):
    """BASELINE full sizes (1280x720 D=128 5-path; 2048x1536 D=256 8-path): properties that do
    not need the (slow) oracle -- determinism, left band invalid, ground-plane disparity law,
    disparity range of valid pixels, keypoints respect the mask, match self-consistency."""
    c = Corridor(name)
    p = c.sgbm_params(mode)
    cam = StereoCamera(c.K(), c.dist(), c.K(), c.dist(), c.rect_params(), p, (c.w, c.h), max_keypoints=nfeat)
    L, R = c.pair(5)
    d1 = cam.stereoSGBM.compute(L, R)
    d2 = cam.stereoSGBM.compute(L, R)
    assert np.array_equal(d1, d2)                                   # idempotent / deterministic
    assert (d1[:, : c.D] == -16).all()                              # columns < numDisparities are INVALID
    valid = d1 >= 0
    assert valid.mean() > 0.5 and d1[valid].max() < c.D * 16
    v = int(c.cy + 0.3 * c.h)
    row = d1[v][d1[v] > 0] / 16.0
    assert abs(np.median(row) - c.B * (v - c.cy) / 1.65) < 0.5      # ground plane: d = B (v - cy) / 1.65
    odo = StereoOdometer(cam, nfeatures=nfeat, preprocessed_frames=True)
    assert odo.update(L, R)
    kps = odo.current_kps
    assert 0.8 * nfeat <= len(kps) <= nfeat + 64
    dd = np.asarray(odo.current_disparity)
    lvl0 = kps.octave == 0
    xy = kps.xy[lvl0].astype(int)
    assert ((dd[xy[:, 1], xy[:, 0]] >= 4) & (dd[xy[:, 1], xy[:, 0]] <= 100)).all()   # mask respected
    idx, dist = cam._ctx.bf_knn2(odo.current_desc, odo.current_desc)
    assert (idx[:, 0] == np.arange(len(kps))).all() and (dist[:, 0] == 0).all()       # self-match


@pytest.mark.parametrize("kw", [dict(rigidity_threshold=0.1, outlier_threshold=0.02), dict(outlier_threshold=0.05),
                                dict(rigidity_threshold=0.02), dict()])
def test_fused_pair_step_equals_public_seams(kw):
    """The one-call device path (vo_pose_pair) and the step-by-step public methods give the same
    decisions and the same pose (tolerance 1e-10: float64 reductions differ in order)."""
    class Generic(StereoOdometer):     # a subclass never takes the fused shortcut
        pass
    c, cam = _rig("C1", max_keypoints=500)
    fused = StereoOdometer(cam, preprocessed_frames=True, **kw)
    plain = Generic(cam, preprocessed_frames=True, **kw)
    for k in range(8):
        L, R = c.pair(k)
        a, b = fused.update(L, R), plain.update(L, R)
        assert a == b and fused.skip_cause == plain.skip_cause and fused.skipped_frames == plain.skipped_frames, k
        assert np.allclose(fused.c_T_w, plain.c_T_w, rtol=0, atol=1e-10), k


def test_c2_full_size_frame_matches_oracle(oracle):
    """One BASELINE-size frame (1280x720, D=128) against the oracle: disparity, keypoints,
    descriptors bit-exact (the oracle needs ~6 s for it)."""
    c = Corridor("C2")
    cam = StereoCamera(c.K(), c.dist(), c.K(), c.dist(), c.rect_params(), c.sgbm_params(), (c.w, c.h), max_keypoints=500)
    L, R = c.pair(9)
    ref16 = oracle.sgbm_compute(L, R, c.sgbm_params(), 0)
    assert np.array_equal(cam.stereoSGBM.compute(L, R), ref16)
    odo = StereoOdometer(cam, preprocessed_frames=True)
    assert odo.update(L, R)
    vr = cam.valid_region_left
    d = (ref16.astype(np.float32) / 16)[vr[1]:vr[3], vr[0]:vr[2]]
    ref = oracle.orb_detect_and_compute(L[vr[1]:vr[3], vr[0]:vr[2]], odo.feature_mask(d), 500)
    assert np.array_equal(odo.current_kps.xy.view(np.uint32), ref["xy"].view(np.uint32))
    assert np.array_equal(odo.current_kps.angle.view(np.uint32), ref["angle"].view(np.uint32))
    assert np.array_equal(odo.current_desc, ref["desc"])


def test_edge_cases_keypoints_and_masks():
    c, cam = _rig("T0", max_keypoints=300)
    odo = StereoOdometer(cam, nfeatures=300, preprocessed_frames=True)
    flat = np.full((c.h, c.w), 90, np.uint8)
    assert odo.update(flat, flat) is False                     # textureless: no disparity, no keypoints
    assert odo.skip_cause == "keypoints" and odo.skipped_frames == 1 and odo.current_img is None
    kps, desc = odo.orb.detectAndCompute(flat, None)
    assert len(kps) == 0 and desc is None                      # cv2 returns ((), None)
    L, R = c.pair(0)
    kps, desc = odo.orb.detectAndCompute(L, np.zeros_like(L))  # everything masked out
    assert len(kps) == 0 and desc is None
    kps, desc = odo.orb.detectAndCompute(L[:50, :50].copy(), None)   # smaller than the 31-px border on each side
    assert len(kps) == 0 and desc is None
    idx, dist = cam._ctx.bf_knn2(np.zeros((3, 32), np.uint8), np.zeros((1, 32), np.uint8))
    assert idx.tolist() == [[0, -1]] * 3                       # fewer than k train rows
    with pytest.raises(IndexError):
        cam._ctx.ratio_filter(idx, dist, 0.8)                  # what m[1] raises in the reference
    # all-inf taps raise ZeroDivisionError like the reference's bilinear_interpolate_pixels
    img = np.full((4, 4, 3), np.inf, np.float32)
    with pytest.raises(ZeroDivisionError):
        odo.bilinear_interpolate_pixels(img, 1.5, 1.5)
    # min_matches larger than the number of keypoints -> "keypoints"; then recovers
    odo2 = StereoOdometer(cam, nfeatures=300, preprocessed_frames=True, min_matches=100000)
    assert odo2.update(L, R) is False and odo2.skip_cause == "keypoints"


def test_context_argument_errors():
    from openvo_amd import _native
    with pytest.raises(_native.VoError):
        _native.Context(0, 32, 32, 16, 64)                     # below the minimum size
    ctx = _native.Context(0, 128, 128, 32, 64)
    with pytest.raises(_native.VoError):
        ctx.set_sgbm(dict(minDisparity=0, numDisparities=40, blockSize=5, P1=8, P2=32, disp12MaxDiff=1, preFilterCap=31,
                          uniquenessRatio=10, speckleWindowSize=0, speckleRange=0))      # not a multiple of 16
    with pytest.raises(_native.VoError):
        ctx.set_sgbm(dict(minDisparity=0, numDisparities=64, blockSize=5, P1=8, P2=32, disp12MaxDiff=1, preFilterCap=31,
                          uniquenessRatio=10, speckleWindowSize=0, speckleRange=0))      # exceeds max_disp
    with pytest.raises(_native.VoError):
        ctx.sgbm_compute_host(np.zeros((256, 256), np.uint8), np.zeros((256, 256), np.uint8))   # exceeds max size
    ctx.close()


@pytest.mark.parametrize("depth", [0, 3, 20])
def test_submitted_host_pairs_equal_plain_updates(depth):
    """SURVEY 8(f) row 3, the ingest step: pairs submitted ahead from host memory (pinned staging, async
    upload, disparity and keypoints on the look-ahead engines) give bit-identical poses to plain
    update(left, right) calls; depth 20 exceeds the frame slots and exercises the host fallback."""
    c, cam = _rig("C1", max_keypoints=500)
    frames = c.pairs(0, 14)
    kw = dict(preprocessed_frames=True, rigidity_threshold=0.1, outlier_threshold=0.02)
    plain = StereoOdometer(cam, **kw)
    want = []
    for L, R in frames:
        ok = plain.update(L, R)
        want.append((ok, plain.c_T_w.copy()))
    odo = StereoOdometer(cam, **kw)
    got = []
    scratch = [np.empty_like(frames[0][0]), np.empty_like(frames[0][1])]

    def feed():   # the caller may reuse its buffers as soon as submit() returns
        for L, R in frames:
            scratch[0][...] = L
            scratch[1][...] = R
            yield scratch[0], scratch[1]

    for ok in odo.run(feed(), depth=depth):
        got.append((ok, odo.c_T_w.copy()))
    assert len(got) == len(want)
    for (a, Ta), (b, Tb) in zip(got, want):
        assert a == b and np.array_equal(Ta, Tb)
    # a consumed handle cannot be used twice
    h = cam.submit(*frames[0], preprocessed=True)
    cam.compute_3d(h, None)
    with pytest.raises(ValueError):
        cam.compute_3d(h, None)


def test_pnp_pose_mode_tracks_the_corridor():
    """Extension (no reference counterpart): pose_method="pnp" runs the RANSAC solvePnP loop on the
    previous frame's 3-D points and the new keypoints; on the synthetic corridor it must follow the
    analytic ground truth about as well as the reference's 3-D/3-D fit does."""
    c, cam = _rig("C1", max_keypoints=500)
    frames = c.pairs(0, 12)
    pnp = StereoOdometer(cam, preprocessed_frames=True, pose_method="pnp")
    ume = StereoOdometer(cam, preprocessed_frames=True, rigidity_threshold=0.1, outlier_threshold=0.02)
    for L, R in frames:
        assert pnp.update(L, R) and ume.update(L, R)
    gt = np.linalg.inv(Corridor.gt_pose(0)) @ Corridor.gt_pose(11)
    e_pnp = np.linalg.norm(pnp.current_pose()[:3, 3] - gt[:3, 3])
    e_ume = np.linalg.norm(ume.current_pose()[:3, 3] - gt[:3, 3])
    assert e_pnp < 0.05 and e_pnp < 3 * e_ume + 0.02
    with pytest.raises(ValueError):
        StereoOdometer(cam, pose_method="icp")


def test_lookahead_keypoints_are_recomputed_when_the_request_differs():
    """Keypoints extracted ahead of time on a look-ahead engine are only handed out for the very
    arguments they were computed with; any other request recomputes and gives what a context without
    look-ahead gives."""
    c, cam = _rig("C1", max_keypoints=500)
    frames = c.pairs(0, 8)
    staged = cam.stage_pairs(frames)
    a = StereoOdometer(cam, nfeatures=500, preprocessed_frames=True)
    for i in range(4):
        assert a.update(staged[i], None)            # frames 4.. are now in flight with 500-feature keypoints
    b = StereoOdometer(cam, nfeatures=300, preprocessed_frames=True)
    x3, disp, left = cam.compute_3d(staged[4], None, preprocessed=True)
    kps300, desc300 = b.orb.detectAndCompute(left, b.feature_mask(disp))
    kps500, desc500 = a.orb.detectAndCompute(left, a.feature_mask(disp))
    c2, cam2 = _rig("C1", max_keypoints=500)
    cam2.lookahead = 0
    x3b, dispb, leftb = cam2.compute_3d(frames[4][0], frames[4][1], preprocessed=True)
    for nf, (k, d) in ((300, (kps300, desc300)), (500, (kps500, desc500))):
        o = StereoOdometer(cam2, nfeatures=nf, preprocessed_frames=True)
        kr, dr = o.orb.detectAndCompute(leftb, o.feature_mask(dispb))
        assert len(k) == len(kr) and np.array_equal(k.xy, kr.xy) and np.array_equal(d, dr)
    assert len(kps300) < len(kps500)


def test_pose_started_ahead_survives_a_change_of_plan():
    """The matching + pose step of the next pairs is started ahead of time when they are already on the
    device; results must not depend on it.  Here the caller breaks the expected order (skips a staged
    pair, then changes a threshold) and every pose must equal the one of an odometer that never looks ahead."""
    c, cam = _rig("C1", max_keypoints=500)
    frames = c.pairs(0, 10)
    kw = dict(preprocessed_frames=True, rigidity_threshold=0.1, outlier_threshold=0.02)
    order = [0, 1, 2, 4, 5, 3, 6, 7, 9]
    c2, cam2 = _rig("C1", max_keypoints=500)
    cam2.lookahead = 0
    ref = StereoOdometer(cam2, **kw)
    want = []
    for j, i in enumerate(order):
        if j == 6:
            ref.match_threshold = 0.75
        want.append((ref.update(*frames[i]), ref.c_T_w.copy()))
    staged = cam.stage_pairs(frames)
    odo = StereoOdometer(cam, **kw)
    for j, i in enumerate(order):
        if j == 6:
            odo.match_threshold = 0.75
        ok = odo.update(staged[i], None)
        assert ok == want[j][0] and np.array_equal(odo.c_T_w, want[j][1]), (j, i)


def test_pose_started_ahead_is_never_reused_for_a_refilled_slot():
    """Dropping the look-ahead (slots get refilled with other pairs) must invalidate pose steps that were
    started ahead of time on the old contents."""
    c, cam = _rig("C1", max_keypoints=500)
    frames = c.pairs(0, 12)
    kw = dict(preprocessed_frames=True, rigidity_threshold=0.1, outlier_threshold=0.02)
    c2, cam2 = _rig("C1", max_keypoints=500)
    cam2.lookahead = 0
    ref = StereoOdometer(cam2, **kw)
    order = [0, 1, 2, 3, 8, 9, 10, 11]          # jump after the look-ahead has been dropped
    want = [(ref.update(*frames[i]), ref.c_T_w.copy()) for i in order]
    staged = cam.stage_pairs(frames)
    odo = StereoOdometer(cam, **kw)
    for j, i in enumerate(order):
        if j == 4:
            cam.reset_lookahead()
        ok = odo.update(staged[i], None)
        assert ok == want[j][0] and np.array_equal(odo.c_T_w, want[j][1]), (j, i)


def test_inflight_lookahead_keypoints_keep_their_own_quotas():
    """Keypoint extractions already queued on the look-ahead engines must not see the per-level quotas of a
    later request with another nfeatures (the level table travels by value in every launch): consume the
    in-flight frames after such a switch and compare with a context that never looks ahead."""
    c, cam = _rig("C1", max_keypoints=500)
    frames = c.pairs(0, 10)
    staged = cam.stage_pairs(frames)
    a = StereoOdometer(cam, nfeatures=500, preprocessed_frames=True)
    for i in range(3):
        assert a.update(staged[i], None)        # frames 3.. are now in flight with 500-feature extractions
    b = StereoOdometer(cam, nfeatures=300, preprocessed_frames=True)
    x3, disp, left = cam.compute_3d(staged[3], None, preprocessed=True)
    k300, _ = b.orb.detectAndCompute(left, b.feature_mask(disp))      # another nfeatures while 4.. are in flight
    got = []
    for i in range(4, 10):
        ok = a.update(staged[i], None)
        got.append((ok, a.current_kps.xy.copy(), np.asarray(a.current_desc).copy(), a.c_T_w.copy()))
    c2, cam2 = _rig("C1", max_keypoints=500)
    cam2.lookahead = 0
    r = StereoOdometer(cam2, nfeatures=500, preprocessed_frames=True)
    for i in (0, 1, 2):
        assert r.update(*frames[i])
    for j, i in enumerate(range(4, 10)):
        ok = r.update(*frames[i])
        assert ok == got[j][0]
        assert np.array_equal(r.current_kps.xy, got[j][1]) and np.array_equal(np.asarray(r.current_desc), got[j][2]), i
        assert np.array_equal(r.c_T_w, got[j][3]), i
    assert 250 <= len(k300) <= 340 and len(got[-1][1]) > 400


def _failed_prefetch_body():
    """Body of test_failed_prefetch_mid_stream_leaves_the_context_usable: runs in a process of its own, against the
    test-only build of the library (libvo355_hooks.so: VO_FAULT_PREFETCH is compiled out of the product)."""
    import os
    from openvo_amd import _native
    c2, cam2 = _rig("C1", max_keypoints=500)
    frames = c2.pairs(0, 9)
    kw = dict(preprocessed_frames=True, rigidity_threshold=0.1, outlier_threshold=0.02)
    ref = StereoOdometer(cam2, **kw)
    want = [(ref.update(L, R), ref.c_T_w.copy()) for L, R in frames]
    os.environ["VO_FAULT_PREFETCH"] = "5"                          # the 5th submission fails after its ingest was enqueued
    c, cam = _rig("C1", max_keypoints=500)
    del os.environ["VO_FAULT_PREFETCH"]
    odo = StereoOdometer(cam, **kw)
    big = np.zeros((c.h + 64, c.w + 64), np.uint8)
    failures = 0

    def submit(pair):
        nonlocal failures
        try:
            return cam.submit(*pair, preprocessed=True)
        except _native.VoError as e:
            assert "injected failure" in str(e)
            failures += 1
            return cam.submit(*pair, preprocessed=True)            # the same pair again: must work now

    it = iter(frames)
    queue = [submit(next(it)) for _ in range(3)]
    got = []
    for k in range(len(frames)):
        if k == 2:
            with pytest.raises(_native.VoError):
                cam.submit(big, big, preprocessed=True)            # VO_E_CAP before any engine is involved
        got.append((odo.update(queue.pop(0), None), odo.c_T_w.copy()))
        nxt = next(it, None)
        if nxt is not None:
            queue.append(submit(nxt))
    assert failures == 1
    for (a, Ta), (b, Tb) in zip(got, want):
        assert a == b and np.array_equal(Ta, Tb)
    assert np.array_equal(cam.stereoSGBM.compute(*frames[3]), cam2.stereoSGBM.compute(*frames[3]))
    print("failed-prefetch body ok")


def test_failed_prefetch_mid_stream_leaves_the_context_usable():
    """A look-ahead submission that fails in the middle of a stream -- before the engine is touched (image larger
    than the context) and INSIDE the engine scope, after its upload was queued (injected through the test-only build
    of the library, libvo355_hooks.so) -- must not leave the context pointing at an engine's stream / workspaces: the
    pairs after it give the poses of a run that never saw a failure."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    hooks = os.path.join(root, "openvo_amd", "libvo355_hooks.so")
    assert os.path.exists(hooks), "build the test-only library first (__graft_entry__.build())"
    env = dict(os.environ, VO355_LIB=hooks)
    r = subprocess.run([sys.executable, "-c", "import tests.test_gpu_pipeline as t; t._failed_prefetch_body()"], cwd=root, env=env,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "failed-prefetch body ok" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]


def _sweep_timeout_body():
    """Body of test_sweep_timeout_raises_and_the_next_pair_is_exact: own process, test-only build of the library.
    VO_FAULT_SWEEP=n makes the n-th diagonal sweep of the context export nothing and give up its waits after 64 polls: the
    real failure path of an oversubscribed GPU (a strip hand-off that never arrives), forced."""
    import os
    from openvo_amd import _native
    kw = dict(preprocessed_frames=True, rigidity_threshold=0.1, outlier_threshold=0.02)
    c0, cam0 = _rig("C1", max_keypoints=500)
    frames = c0.pairs(0, 10)
    ref = StereoOdometer(cam0, **kw)
    want = [(ref.update(L, R), ref.c_T_w.copy()) for L, R in frames]
    assert cam0._ctx.sgbm_sweep_status() == 0

    # (a) plain update() calls on host arrays: the 3rd pair's sweep fails
    os.environ["VO_FAULT_SWEEP"] = "3"
    c, cam = _rig("C1", max_keypoints=500)
    odo = StereoOdometer(cam, **kw)
    got, raised = [], 0
    for k, (L, R) in enumerate(frames[:6]):
        try:
            ok = odo.update(L, R)
        except _native.SweepTimeout as e:
            assert e.code == _native.VO_E_SWEEP and k == 2, (k, str(e))
            raised += 1
            before = (odo.skipped_frames, odo.c_T_w.copy(), odo.current_kps)
            ok = odo.update(L, R)                                  # the same pair again: exact now
            assert before[0] == 0 and odo.current_kps is not before[2]
        got.append((ok, odo.c_T_w.copy()))
    assert raised == 1 and cam._ctx.sgbm_sweep_status() == 1
    for (a, Ta), (b, Tb) in zip(got, want):
        assert a == b and np.array_equal(Ta, Tb)
    # the seam itself refuses the disparity too, and the one after it is exact again
    os.environ["VO_FAULT_SWEEP"] = "1"
    c3, cam3 = _rig("C1", max_keypoints=500)
    with pytest.raises(_native.SweepTimeout):
        cam3.stereoSGBM.compute(*frames[1])
    assert np.array_equal(cam3.stereoSGBM.compute(*frames[1]), cam0.stereoSGBM.compute(*frames[1]))

    # (b) staged pairs with the default look-ahead (pose steps begun ahead included): the 5th sweep = staged pair 4 fails while
    # it runs AHEAD on an engine; update(4) raises -- not update(3), which only looked at it -- and passing pair 4 again works
    os.environ["VO_FAULT_SWEEP"] = "5"
    c2, cam2 = _rig("C1", max_keypoints=500)
    del os.environ["VO_FAULT_SWEEP"]
    staged = cam2.stage_pairs(frames)
    odo2 = StereoOdometer(cam2, **kw)
    got2, raised2 = [], []
    for k in range(len(frames)):
        try:
            ok = odo2.update(staged[k], None)
        except _native.SweepTimeout:
            raised2.append(k)
            ok = odo2.update(staged[k], None)
        got2.append((ok, odo2.c_T_w.copy()))
    assert raised2 == [4], raised2
    assert cam2._ctx.sgbm_sweep_status() == 1
    for (a, Ta), (b, Tb) in zip(got2, want):
        assert a == b and np.array_equal(Ta, Tb)

    # (c) the same through run() (host pairs submitted ahead): by default the exception ends the generator and EVERY slot the
    # pairs submitted ahead held is given back (they used to stay reserved: one timeout stranded up to 25 of the 28 slots);
    # a second run() on the same camera is bit-exact.  With on_sweep_timeout="skip" the generator yields None and carries on.
    from openvo_amd.stereo_camera import _RESERVED
    os.environ["VO_FAULT_SWEEP"] = "4"
    c4, cam4 = _rig("C1", max_keypoints=500)
    del os.environ["VO_FAULT_SWEEP"]
    odo4 = StereoOdometer(cam4, **kw)
    got4 = []
    with pytest.raises(_native.SweepTimeout):
        for ok in odo4.run(iter(frames)):
            got4.append((ok, odo4.c_T_w.copy()))
    assert len(got4) == 3                                        # pairs 0..2 yielded, pair 3 raised
    assert not any(o is _RESERVED for o in cam4._slot_owner), "slots of pairs submitted ahead stayed reserved"
    assert cam4._ctx.lookahead_depth() == 0
    for (a, Ta), (b, Tb) in zip(got4, want):
        assert a == b and np.array_equal(Ta, Tb)
    odo4b = StereoOdometer(cam4, **kw)                           # a fresh odometer on the SAME camera: all ten pairs, exact
    got4b = [(ok, odo4b.c_T_w.copy()) for ok in odo4b.run(iter(frames))]
    assert len(got4b) == len(want) and not any(o is _RESERVED for o in cam4._slot_owner)
    for (a, Ta), (b, Tb) in zip(got4b, want):
        assert a == b and np.array_equal(Ta, Tb)
    os.environ["VO_FAULT_SWEEP"] = "4"
    c5, cam5 = _rig("C1", max_keypoints=500)
    del os.environ["VO_FAULT_SWEEP"]
    odo5 = StereoOdometer(cam5, **kw)
    got5 = list(odo5.run(iter(frames), on_sweep_timeout="skip"))
    assert got5.count(None) == 1 and got5[3] is None and len(got5) == len(frames)
    assert not any(o is _RESERVED for o in cam5._slot_owner)
    # a generator closed early gives its slots back too
    g = odo5.run(iter(frames))
    next(g)
    g.close()
    assert not any(o is _RESERVED for o in cam5._slot_owner) and cam5._ctx.lookahead_depth() == 0
    print("sweep-timeout body ok")


def test_sweep_timeout_raises_and_the_next_pair_is_exact():
    """A strip hand-off of the diagonal aggregation sweep that gives up waiting leaves that pair's disparity undefined.
    The product must say so -- update() raises SweepTimeout (VO_E_SWEEP), never returns a pose from it -- and the failure
    must not outlive the pair: the launch's abort word is cleared by the workspace's next run, so the same pair submitted
    again (and every later one) is bit-exact.  Forced through the test-only build (libvo355_hooks.so, VO_FAULT_SWEEP)."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    hooks = os.path.join(root, "openvo_amd", "libvo355_hooks.so")
    assert os.path.exists(hooks), "build the test-only library first (__graft_entry__.build())"
    env = dict(os.environ, VO355_LIB=hooks)
    r = subprocess.run([sys.executable, "-c", "import tests.test_gpu_pipeline as t; t._sweep_timeout_body()"], cwd=root, env=env,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "sweep-timeout body ok" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]


def test_product_library_has_no_failure_injection(monkeypatch):
    """The product build ignores VO_FAULT_PREFETCH / VO_FAULT_SWEEP: nothing a user's environment holds can make a submission
    or a sweep fail."""
    monkeypatch.setenv("VO_FAULT_PREFETCH", "1")
    monkeypatch.setenv("VO_FAULT_SWEEP", "1")
    c, cam = _rig("T0", max_keypoints=300)
    monkeypatch.delenv("VO_FAULT_PREFETCH")
    monkeypatch.delenv("VO_FAULT_SWEEP")
    sp = [cam.submit(*c.pair(k), preprocessed=True) for k in range(3)]
    odo = StereoOdometer(cam, nfeatures=300, preprocessed_frames=True)
    assert odo.update(sp[0], None) is True


def test_copy_ceiling_probe_is_sane_and_leaves_the_context_usable(oracle):
    """vo_measure_copy (the bench's streaming-copy ceiling) returns a plausible rate and, although it scribbles over the
    cost volume, the next SGBM run is unaffected (every run rebuilds the volumes)."""
    c = Corridor("C1")
    cam = StereoCamera(c.K(), c.dist(), c.K(), c.dist(), c.rect_params(), c.sgbm_params(), (c.w, c.h), max_keypoints=500)
    L, R = c.pair(1)
    before = cam.stereoSGBM.compute(L, R)
    for nt in (False, True):
        rate = cam._ctx.measure_copy(0, 5, nt)
        assert 100.0 < rate < 40000.0, rate                      # GB/s, read + written (HBM peak 8000; a volume this small can sit in the 256 MB MALL)
    assert np.array_equal(cam.stereoSGBM.compute(L, R), before)
    with pytest.raises(Exception):
        cam._ctx.measure_copy(0, 0, False)                       # reps must be positive


def test_host_staging_thread_begin_wait_fetch_and_prefetch():
    """vo_host_stage_begin hands the copy into pinned staging to the library's own thread: a buffer read back (fetch waits for
    the copy) holds the pair; a pair prefetched straight after begin (no explicit wait) gives the disparity of the plain host
    path; a buffer refilled while its previous upload may still be running (the staging thread waits for that upload) does
    too; wait on an idle buffer returns at once; bad arguments are refused."""
    c, cam = _rig("C1", max_keypoints=500)
    ctx = cam._ctx
    pairs = c.pairs(0, 6)
    want = [cam.stereoSGBM.compute(L, R) for L, R in pairs]
    w, h, ch, keep = ctx.host_stage_begin(2, *pairs[0])
    assert (w, h, ch) == (c.w, c.h, 1)
    Lb, Rb = ctx.host_stage_fetch(2, w, h, ch)
    assert np.array_equal(Lb, pairs[0][0]) and np.array_equal(Rb, pairs[0][1])
    ctx.host_stage_wait(2)
    ctx.host_stage_wait(7)                                            # never used: nothing pending
    for k, (L, R) in enumerate(pairs):                                # one buffer over and over: every refill meets the last upload
        w, h, ch, keep = ctx.host_stage_begin(5, L, R)
        ctx.prefetch_host_staged(3 + (k & 1), 5, w, h, ch, True)
        assert np.array_equal(ctx.download_left(3 + (k & 1), L.shape), L), k
        assert np.array_equal(ctx.download_disparity_f32(3 + (k & 1), L.shape), want[k].astype(np.float32) / 16.0), k
    with pytest.raises(Exception):
        ctx.host_stage_begin(_native.VO_NUM_HOST_STAGE, *pairs[0])
    with pytest.raises(Exception):
        ctx.host_stage_wait(-1)


@pytest.mark.parametrize("env", [{"VO_STAGGER": "0"}, {"VO_STAGGER": "2", "VO_ENGINES": "5"}, {"VO_POSE_AHEAD": "0"},
                                 {"VO_POSE_AHEAD": "7", "VO_LOOKAHEAD": "9"}])
def test_scheduling_knobs_change_nothing_but_the_order(env, monkeypatch):
    """The staggered start of the look-ahead engines (VO_STAGGER), the number of engines and how many pose steps are begun
    ahead (VO_POSE_AHEAD) only order work on the device: a staged stream gives bit-identical flags and poses under any of them."""
    c = Corridor("C1")
    frames = c.pairs(0, 14)
    kw = dict(preprocessed_frames=True, rigidity_threshold=0.1, outlier_threshold=0.02)

    def chain():
        cam = StereoCamera(c.K(), c.dist(), c.K(), c.dist(), c.rect_params(), c.sgbm_params(), (c.w, c.h), max_keypoints=500)
        odo = StereoOdometer(cam, **kw)
        staged = cam.stage_pairs(frames)
        out = [(odo.update(s, None), odo.c_T_w.copy()) for s in staged]
        cam._ctx.close()
        return out

    want = chain()
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    got = chain()
    assert sum(a for a, _ in want) >= 12
    for (a, Ta), (b, Tb) in zip(got, want):
        assert a == b and np.array_equal(Ta, Tb)


def test_engine_count_as_a_constructor_argument_bounds_the_footprint_and_changes_no_result():
    """StereoCamera(..., engines=2): the context creates at most two look-ahead workspaces (device memory in use stays below
    the default's), the look-ahead follows, lowering it later is allowed, and flags and poses stay bit-identical."""
    c = Corridor("C1")
    frames = c.pairs(0, 12)
    kw = dict(preprocessed_frames=True, rigidity_threshold=0.1, outlier_threshold=0.02)

    def chain(**cam_kw):
        cam = StereoCamera(c.K(), c.dist(), c.K(), c.dist(), c.rect_params(), c.sgbm_params(), (c.w, c.h), max_keypoints=500, **cam_kw)
        odo = StereoOdometer(cam, **kw)
        staged = cam.stage_pairs(frames)
        out = []
        for k, s in enumerate(staged):
            if k == 6 and cam_kw.get("engines") == 3:
                assert cam._ctx.set_engines(1) == 1                      # lowered mid-stream: later pairs all go to engine 0
            out.append((odo.update(s, None), odo.c_T_w.copy()))
        used = cam._ctx.mem_used() if hasattr(cam._ctx, "mem_used") else None
        n = cam._ctx.set_engines(0)
        la = cam.lookahead
        cam._ctx.close()
        return out, n, la, used

    want, n_def, la_def, _ = chain()
    assert n_def >= 2 and la_def >= 2
    for eng in (2, 3):
        got, n, la, _ = chain(engines=eng)
        assert n == (1 if eng == 3 else 2) and la <= eng + 8
        for (a, Ta), (b, Tb) in zip(got, want):
            assert a == b and np.array_equal(Ta, Tb)
    with pytest.raises(ValueError):
        chain(engines=0)
    cam = StereoCamera(c.K(), c.dist(), c.K(), c.dist(), c.rect_params(), c.sgbm_params(), (c.w, c.h), engines=1000, lookahead=99)
    assert 2 <= cam._ctx.set_engines(0) <= 24 and cam.lookahead == _native.VO_NUM_SLOTS - 3
    cam._ctx.close()


def test_staged_gray_pairs_narrower_than_the_disparity_range_keep_their_own_images():
    """Width <= numDisparities: every pixel is INVALID and no cost kernel runs -- but on the in-place ingest path of staged
    rectified gray pairs it is the first SGBM kernel that leaves the slot's copy of the pair behind, so the early return must
    copy the images itself (it used to leave the PREVIOUS pair's images in the slot: ADVICE round 4)."""
    c = Corridor("T0")
    p = dict(c.sgbm_params(), numDisparities=128)
    w = 96
    cam = StereoCamera(c.K(), c.dist(), c.K(), c.dist(), c.rect_params(), p, (w, c.h), max_keypoints=300)
    frames = [(np.ascontiguousarray(L[:, :w]), np.ascontiguousarray(R[:, :w])) for L, R in c.pairs(0, 5)]
    staged = cam.stage_pairs(frames)
    vr = cam.valid_region_left
    for k in (0, 1, 2, 3, 4, 1):
        xyz, disp, left = cam.compute_3d(staged[k], None, preprocessed=True)
        assert np.array_equal(np.asarray(left), frames[k][0][vr[1]:vr[3], vr[0]:vr[2]]), k
        assert (np.asarray(disp) == -1.0).all()                      # (minDisparity - 1) everywhere
    cam._ctx.close()


def test_lookahead_depth_counts_what_is_really_in_flight():
    """vo_lookahead_depth: pairs submitted ahead and neither consumed nor dropped.  A reset of the look-ahead (and a fresh
    stage_pairs) must bring it back to zero although the dropped slots' work may still be running."""
    c, cam = _rig("C1", max_keypoints=500)
    odo = StereoOdometer(cam, rigidity_threshold=0.1, outlier_threshold=0.02, preprocessed_frames=True)
    staged = cam.stage_pairs(c.pairs(0, 12))
    ctx = cam._ctx
    assert ctx.lookahead_depth() == 0
    assert odo.update(staged[0], None)
    ahead = min(int(cam.lookahead), 11)
    assert 0 < ctx.lookahead_depth() <= ahead                    # pairs 1.. were started behind pair 0 (the odometer may already
    assert odo.update(staged[1], None)                           # have waited for the next ones' keypoints)
    assert 0 < ctx.lookahead_depth() <= ahead
    cam.reset_lookahead()
    assert ctx.lookahead_depth() == 0
    assert odo.update(staged[5], None) in (True, False)          # an index nobody predicted after the reset: recomputed
    cam.stage_pairs(c.pairs(0, 4))
    assert ctx.lookahead_depth() == 0
    assert ctx.sgbm_sweep_status() == 0


@pytest.mark.parametrize("seed", [21, 22])
def test_update_sequence_fuzz_against_the_oracle_odometer(seed):
    """Random odometers (features 50 .. 300, ratio 0.5 .. 0.95, rigidity 0 .. 5, outlier 0 .. 0.5, min_matches 3 .. 100) on a
    14-frame sequence in which frames are replaced at random by textureless ones ("keypoints"), by jumps to another frame of
    the sequence ("bigdist" / "rigidity" / "matches"), by pairs with a broken right image (few valid disparities) and by
    posterised ones: every update()'s result, skip_cause, skipped_frames and chained pose equal the oracle odometer's -- the
    fused device path, the one-frame-back fallback and the speculative pose steps included."""
    from oracle.odometer import RefStereoCamera, RefStereoOdometer
    rng = np.random.default_rng(seed)
    c = Corridor("T0")
    base = c.pairs(0, 14)
    causes = set()
    for trial in range(5):
        nfeat = int(rng.choice([50, 150, 300]))
        kw = dict(nfeatures=nfeat, match_threshold=float(rng.choice([0.5, 0.8, 0.95])), rigidity_threshold=float(rng.choice([0, 0, 0.02, 0.1, 5.0])),
                  outlier_threshold=float(rng.choice([0, 0, 0.005, 0.02, 0.5])), min_matches=int(rng.choice([3, 10, 40, 100])))
        cam = StereoCamera(c.K(), c.dist(), c.K(), c.dist(), c.rect_params(), c.sgbm_params(), (c.w, c.h), max_keypoints=nfeat)
        rcam = RefStereoCamera(cam.Q, cam.valid_region_left, c.sgbm_params())
        odo = StereoOdometer(cam, preprocessed_frames=True, **kw)
        rodo = RefStereoOdometer(rcam, preprocessed_frames=True, **kw)
        for k in range(14):
            L, R = base[k]
            r = rng.random()
            if r < 0.12:
                L = np.full_like(L, 90); R = np.full_like(R, 90)
            elif r < 0.22:
                L, R = base[int(rng.integers(0, 14))]
            elif r < 0.30:
                R = np.ascontiguousarray(R[::-1])
            elif r < 0.36:
                L = (L.astype(np.int32) // 8 * 8).astype(np.uint8)
            a, b = odo.update(L, R), rodo.update(L, R)
            assert a == b and odo.skip_cause == rodo.skip_cause and odo.skipped_frames == rodo.skipped_frames, (trial, k, kw)
            assert np.allclose(odo.c_T_w, rodo.c_T_w, rtol=0, atol=1e-8), (trial, k, kw)
            if not a:
                causes.add(odo.skip_cause)
    assert len(causes) >= 2                                # the disturbances do exercise the rejection paths


@pytest.mark.parametrize("seed", [51, 52])
def test_hand_over_fuzz_every_way_of_passing_a_pair_equals_plain_updates(seed):
    """A 30-pair sequence (consecutive frames, now and then a textureless pair or a repeated one) is handed to one odometer in
    random chunks, each in a random way -- plain update(L, R); run() at depth 0 / 1 / 3 / 20; pairs staged in HBM and consumed in
    order; pairs submitted ahead; staged pairs abandoned half way (reset_lookahead, the look-ahead work voided) and finished with
    plain calls; a foreign compute_3d on the same camera in between -- under a random look-ahead depth (0 / 2 / 7 / 18 / 24): every
    update()'s result, skip_cause, skipped_frames and pose are bit-identical to plain updates on a camera without look-ahead."""
    rng = np.random.default_rng(seed)
    c = Corridor("T0")
    frames = c.pairs(0, 40)
    flat = (np.full_like(frames[0][0], 90), np.full_like(frames[0][1], 90))
    kw = dict(rigidity_threshold=0.1, outlier_threshold=0.02, preprocessed_frames=True)
    for trial in range(4):
        seq, k = [], 0
        while len(seq) < 30:
            r = rng.random()
            if r < 0.08:
                seq.append(flat)
            elif r < 0.14 and seq:
                seq.append(seq[-1])
            else:
                seq.append(frames[k % 40]); k += 1
        cam0 = StereoCamera(c.K(), c.dist(), c.K(), c.dist(), c.rect_params(), c.sgbm_params(), (c.w, c.h), max_keypoints=200)
        cam0.lookahead = 0
        ref = StereoOdometer(cam0, nfeatures=200, **kw)
        want = []
        for L, R in seq:
            ok = ref.update(L, R)
            want.append((ok, ref.skip_cause, ref.skipped_frames, ref.c_T_w.copy()))
        cam = StereoCamera(c.K(), c.dist(), c.K(), c.dist(), c.rect_params(), c.sgbm_params(), (c.w, c.h), max_keypoints=200)
        cam.lookahead = int(rng.choice([0, 2, 7, 18, 24]))
        odo = StereoOdometer(cam, nfeatures=200, **kw)
        got, log, i = [], [], 0
        rec = lambda ok: got.append((ok, odo.skip_cause, odo.skipped_frames, odo.c_T_w.copy()))
        while i < len(seq):
            op, chunk = int(rng.integers(0, 6)), seq[i:i + int(rng.integers(1, 7))]
            log.append((op, len(chunk)))
            if op == 0:
                for L, R in chunk:
                    rec(odo.update(L, R))
            elif op == 1:
                for ok in odo.run(iter(chunk), depth=int(rng.choice([0, 1, 3, 20]))):
                    rec(ok)
            elif op == 2:
                for s in cam.stage_pairs(chunk):
                    rec(odo.update(s, None))
            elif op == 3:
                for s in [cam.submit(L, R, preprocessed=True) for L, R in chunk]:
                    rec(odo.update(s, None))
            elif op == 4:
                st, half = cam.stage_pairs(chunk), max(1, len(chunk) // 2)
                for s in st[:half]:
                    rec(odo.update(s, None))
                odo.reset_lookahead()
                for L, R in chunk[half:]:
                    rec(odo.update(L, R))
            else:
                cam.compute_3d(*frames[int(rng.integers(0, 40))], preprocessed=True)
                for L, R in chunk:
                    rec(odo.update(L, R))
            i += len(chunk)
        assert len(got) == len(want)
        for j, (a, b) in enumerate(zip(got, want)):
            assert a[:3] == b[:3] and np.array_equal(a[3], b[3]), (trial, j, a[:3], b[:3], cam.lookahead, log)


def test_abi_misuse_returns_a_status_and_never_crashes():
    """Every context-taking entry point with a NULL context, NULL data pointers, hostile integers (-1, INT_MIN, INT_MAX, slot
    28 ...) and in the wrong order on a fresh context: a status code every time, never a fault -- in a process of its own so
    that a crash would be this test's failure, not the runner's end.  Only harmless calls may report success."""
    import os, subprocess, sys
    script = os.path.join(os.path.dirname(__file__), "abi_misuse.py")
    r = subprocess.run([sys.executable, script], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, (r.returncode, r.stdout[-500:], r.stderr[-1500:])
    last = r.stdout.strip().splitlines()[-1].split()
    assert last[0] == "survived" and int(last[1]) > 300
    harmless = {"vo_bf_knn2_hamming", "vo_enable_timing", "vo_get_timings", "vo_host_stage_wait", "vo_lookahead_drop", "vo_set_lookahead_orb",
                "vo_set_roi", "vo_sgbm_last_geometry", "vo_synchronize", "vo_lookahead_depth", "vo_sgbm_last_schedule", "vo_sgbm_sweep_status",
                "vo_slot_ready", "vo_sgbm_sweep_stats", "vo_stage_pairs_alloc"}
    assert set(last[2:]) <= harmless, set(last[2:]) - harmless


def test_random_rigs_unrectified_input_through_update_and_run_against_the_oracle():
    """Six random rigs (lens distortion, shifted principal point, a relative rotation and a baseline that is not along x: the
    valid ROI then starts off the origin and the rectification maps matter), the reference's default input (unrectified pairs,
    gray or BGR), through plain update() calls or run(depth=3): every result, skip_cause, skipped_frames and chained pose
    against the oracle odometer working through the same maps (cvtColor -> remap -> SGBM -> crop -> ORB -> pose)."""
    from openvo_amd import calib
    from oracle.odometer import RefStereoCamera, RefStereoOdometer
    rng = np.random.default_rng(71)
    c = Corridor("T0")
    frames = c.pairs(0, 8)
    done = 0
    for trial in range(6):
        dist = np.array([rng.uniform(-0.2, 0.1), rng.uniform(-0.03, 0.05), rng.uniform(-0.002, 0.002), rng.uniform(-0.002, 0.002), rng.uniform(-0.01, 0.01)])
        if trial % 3 == 0:
            dist[:] = 0
        K2 = c.K().copy()
        K2[0, 2] += rng.uniform(-4, 4); K2[1, 2] += rng.uniform(-4, 4)
        rect = {"R": calib.rodrigues_vec_to_mat(rng.uniform(-0.01, 0.01, 3)), "T": np.array([-c.B, rng.uniform(-0.003, 0.003), rng.uniform(-0.003, 0.003)])}
        cam = StereoCamera(c.K(), dist, K2, dist * rng.uniform(0.8, 1.2), rect, c.sgbm_params(), (c.w, c.h), max_keypoints=200)
        vr = cam.valid_region_left
        if vr[2] - vr[0] < 80 or vr[3] - vr[1] < 80:
            continue
        rcam = RefStereoCamera(cam.Q, vr, c.sgbm_params(), maps=((cam.map_left_1, cam.map_left_2), (cam.map_right_1, cam.map_right_2)))
        kw = dict(nfeatures=200, rigidity_threshold=float(rng.choice([0, 0.1])), outlier_threshold=float(rng.choice([0, 0.02])))
        odo, rodo = StereoOdometer(cam, **kw), RefStereoOdometer(rcam, **kw)
        colour, via_run = rng.random() < 0.5, rng.random() < 0.5
        seq = [(np.ascontiguousarray(np.stack([L, L, L], -1)), np.ascontiguousarray(np.stack([R, R, R], -1))) if colour else (L, R) for L, R in frames]
        got = []
        if via_run:
            for ok in odo.run(iter(seq), depth=3):
                got.append((ok, odo.skip_cause, odo.skipped_frames, odo.c_T_w.copy()))
        else:
            for L, R in seq:
                got.append((odo.update(L, R), odo.skip_cause, odo.skipped_frames, odo.c_T_w.copy()))
        for k, (L, R) in enumerate(seq):
            b = rodo.update(L, R)
            a = got[k]
            assert a[0] == b and a[1] == rodo.skip_cause and a[2] == rodo.skipped_frames, (trial, k, vr, colour, via_run)
            assert np.allclose(a[3], rodo.c_T_w, rtol=0, atol=1e-8), (trial, k, vr)
        done += 1
    assert done >= 4
