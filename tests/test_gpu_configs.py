"""Every BASELINE.json configuration at ITS OWN size, HIP path (through the C ABI) against the CPU oracle:

  C2  1280x720, D=128, 5-path: a 4-pair pose chain with the reference's default odometer parameters AND with
      the thresholds bench.py uses (every frame: disparity, keypoints, descriptors bit-exact; chained pose 1e-9)
  C4  2048x1536, D=256, MODE_HH: a full frame against oracle.sgbm_compute and a two-frame pose step
  C5  mono 1920x1080: ORB 8000, 8000 x 8000 Hamming kNN + ratio, 5000-iteration essential-matrix RANSAC and
      5000-iteration solvePnP RANSAC (no openVO counterpart: parity vs the build's own restatement only)
  C3  (C2 sharded over 8 GPUs) needs an 8-GPU node; its host logic is covered in tests/test_sharding.py.

The oracle needs 2 s (C2) to 25 s (C4) per frame, so this file takes a few minutes."""
import numpy as np
import pytest

from openvo_amd import StereoCamera, StereoOdometer, _native
from openvo_amd.synth import Corridor

pytestmark = pytest.mark.gpu


class _CachedRefCamera:
    """RefStereoCamera whose (slow) per-frame result is shared by several oracle odometers."""

    def __init__(self, rcam):
        self.rcam, self.cache, self.disp16 = rcam, {}, {}
        self.Q, self.valid_region_left = rcam.Q, rcam.valid_region_left

    def compute_3d(self, L, R, preprocessed=False):
        key = (L.ctypes.data, R.ctypes.data)
        if key not in self.cache:
            self.cache[key] = self.rcam.compute_3d(L, R, preprocessed=preprocessed)
            self.disp16[key] = self.rcam.last_disp16
        self.last_disp16 = self.disp16[key]
        return self.cache[key]


def _check_frame(odo, rodo, rcam, cam, k):
    vr = cam.valid_region_left
    d16 = np.rint(np.asarray(odo.current_disparity) * 16).astype(np.int16)
    assert np.array_equal(d16, rcam.last_disp16[vr[1]:vr[3], vr[0]:vr[2]]), "frame %d: disparity" % k
    assert np.array_equal(odo.current_kps.xy.view(np.uint32), rodo.cur["kps"]["xy"].view(np.uint32)), "frame %d: keypoints" % k
    assert np.array_equal(odo.current_kps.octave, rodo.cur["kps"]["octave"])
    assert np.array_equal(odo.current_kps.angle.view(np.uint32), rodo.cur["kps"]["angle"].view(np.uint32))
    assert np.array_equal(odo.current_desc, rodo.cur["desc"]), "frame %d: descriptors" % k


def test_c2_pose_chain_default_and_bench_parameters():
    """BASELINE config 2: five 1280x720 frames = four pose steps, once with the reference's defaults
    (rigidity_threshold = outlier_threshold = 0, stereo_odometer.py:14-15) and once with the thresholds of the
    headline bench; both against the oracle odometer on the same frames."""
    from oracle.odometer import RefStereoCamera, RefStereoOdometer
    c = Corridor("C2")
    cam = StereoCamera(c.K(), c.dist(), c.K(), c.dist(), c.rect_params(), c.sgbm_params(), (c.w, c.h), max_keypoints=500)
    rcam = _CachedRefCamera(RefStereoCamera(cam.Q, cam.valid_region_left, c.sgbm_params()))
    frames = c.pairs(20, 5)
    for kw in (dict(), dict(rigidity_threshold=0.1, outlier_threshold=0.02)):
        odo = StereoOdometer(cam, preprocessed_frames=True, **kw)
        rodo = RefStereoOdometer(rcam, preprocessed_frames=True, **kw)
        for k, (L, R) in enumerate(frames):
            a, b = odo.update(L, R), rodo.update(L, R)
            assert a == b and odo.skip_cause == rodo.skip_cause and odo.skipped_frames == rodo.skipped_frames, (kw, k)
            if a:
                _check_frame(odo, rodo, rcam, cam, k)
            assert np.allclose(odo.c_T_w, rodo.c_T_w, rtol=0, atol=1e-9), (kw, k)
        ate = np.linalg.norm(odo.current_pose()[:3, 3] - rodo.current_pose()[:3, 3])
        assert ate <= 1e-4                                     # BASELINE target: pose ATE <= 1e-4 vs the CPU path
        if kw:
            gt = np.linalg.inv(Corridor.gt_pose(20)) @ Corridor.gt_pose(24)
            assert np.linalg.norm(odo.current_pose()[:3, 3] - gt[:3, 3]) < 0.05   # and it does track the corridor
    assert cam._ctx.sgbm_sweep_status() == 0 and cam._ctx.sgbm_last_schedule() == _native.SCHED_DIAG


def test_c4_full_frame_and_pose_step():
    """BASELINE config 4: 2048x1536, 256 disparities, 8-path MODE_HH (W + E volume, reverse + forward diagonal sweep) --
    two full frames against the oracle (disparity, keypoints, descriptors bit-exact) and the pose step between them."""
    from oracle.odometer import RefStereoCamera, RefStereoOdometer
    c = Corridor("C4")
    p = c.sgbm_params(mode=1)
    cam = StereoCamera(c.K(), c.dist(), c.K(), c.dist(), c.rect_params(), p, (c.w, c.h), max_keypoints=500)
    rcam = RefStereoCamera(cam.Q, cam.valid_region_left, p, mode=1)
    kw = dict(preprocessed_frames=True, rigidity_threshold=0.1, outlier_threshold=0.02)
    odo, rodo = StereoOdometer(cam, **kw), RefStereoOdometer(rcam, **kw)
    for k in (3, 4):
        L, R = c.pair(k)
        a, b = odo.update(L, R), rodo.update(L, R)
        assert a and b, (k, odo.skip_cause, rodo.skip_cause)
        _check_frame(odo, rodo, rcam, cam, k)
        if k == 3:
            full = cam.stereoSGBM.compute(L, R)                  # the cv2-object seam on the same frame
            assert np.array_equal(full, rcam.last_disp16)
            assert (full[:, :c.D] == -16).all() and (full >= 0).mean() > 0.5
    assert np.allclose(odo.c_T_w, rodo.c_T_w, rtol=0, atol=1e-9)
    assert np.linalg.norm(odo.c_T_w[:3, 3]) > 0.1                # the camera did move between the two frames
    assert cam._ctx.sgbm_sweep_status() == 0 and cam._ctx.sgbm_last_schedule() == _native.SCHED_DIAG


@pytest.fixture(scope="module")
def c5():
    c = Corridor("C5")
    ctx = _native.Context(0, c.w, c.h, 16, 8000)
    frames = [c.pair(k)[0] for k in (0, 1)]
    yield c, ctx, frames
    ctx.close()


def test_c5_orb_8000_knn_and_essential_ransac(oracle, c5):
    """BASELINE config 5 front half: ORB with nfeatures = 8000 on two 1920x1080 frames, the ~8000 x 8000
    brute-force Hamming kNN-2 + ratio test between them, and 5000 essential-matrix hypotheses scored on the
    surviving correspondences.  Keypoint sets, descriptors, match indices and distances, every hypothesis'
    inlier count, the winner and its mask bit-exact; E to 1e-12."""
    c, ctx, frames = c5
    got = [ctx.orb_host(f, None, 8000) for f in frames]
    ref = [oracle.orb_detect_and_compute(f, None, 8000) for f in frames]
    for g, r in zip(got, ref):
        assert 7000 <= len(r["xy"]) <= 8400                      # ties may exceed nfeatures (retainBest)
        assert np.array_equal(g["octave"], r["octave"]) and np.array_equal(g["xy"].view(np.uint32), r["xy"].view(np.uint32))
        assert np.array_equal(g["response"].view(np.uint32), r["response"].view(np.uint32))
        assert np.array_equal(g["angle"].view(np.uint32), r["angle"].view(np.uint32))
        assert np.array_equal(g["desc"], r["desc"])
    gi, gd = ctx.bf_knn2(got[0]["desc"], got[1]["desc"])
    ri, rd = oracle.bf_knn2_hamming(ref[0]["desc"], ref[1]["desc"])
    assert np.array_equal(gi, ri) and np.array_equal(gd, rd)
    q, t = ctx.ratio_filter(gi, gd, 0.8)
    rq, rt = oracle.ratio_filter(ri, rd, 0.8)
    assert np.array_equal(q, rq) and np.array_equal(t, rt) and len(q) > 2000
    p1, p2 = got[0]["xy"][q], got[1]["xy"][t]
    K4 = [c.f, c.f, c.cx, c.cy]
    rr = oracle.ransac_essential(p1, p2, K4, 5000, 1.0, 4321)
    gr = ctx.ransac_essential(p1, p2, K4, 5000, 1.0, 4321, want_counts=True)
    assert np.array_equal(gr["counts"], rr["counts"]) and gr["best_iter"] == rr["best_iter"] and gr["best_count"] == rr["best_count"]
    assert np.array_equal(gr["mask"], rr["mask"]) and np.allclose(gr["E"], rr["E"], rtol=0, atol=1e-12)
    assert gr["best_count"] > 0.5 * len(q)                       # the corridor motion is recovered from the matches


def test_c5_pnp_ransac_8000_points_5000_hypotheses(oracle, c5):
    """BASELINE config 5 sizes through the solvePnP scorer (north_star's "RANSAC essential-matrix / solvePnP
    hypothesis-scoring loop"): 8000 correspondences x 5000 hypotheses = 40 M residual evaluations."""
    c, ctx, _ = c5
    rng = np.random.default_rng(11)
    n = 8000
    X = np.stack([rng.uniform(-8, 8, n), rng.uniform(-3, 3, n), rng.uniform(4, 40, n)], 1).astype(np.float32)
    Xc = X.astype(np.float64) + np.array([0.05, -0.02, -0.3])
    uv = np.stack([c.f * Xc[:, 0] / Xc[:, 2] + c.cx, c.f * Xc[:, 1] / Xc[:, 2] + c.cy], 1) + rng.normal(0, 0.3, (n, 2))
    uv[: n * 3 // 10] += rng.uniform(-60, 60, (n * 3 // 10, 2))
    uv = uv.astype(np.float32)
    K4 = [c.f, c.f, c.cx, c.cy]
    rr = oracle.ransac_pnp(X, uv, K4, 5000, 2.0, 4321)
    gr = ctx.ransac_pnp(X, uv, K4, 5000, 2.0, 4321, want_counts=True)
    assert np.array_equal(gr["counts"], rr["counts"]) and gr["best_iter"] == rr["best_iter"] and gr["best_count"] == rr["best_count"]
    assert np.array_equal(gr["mask"], rr["mask"]) and np.allclose(gr["Rt"], rr["Rt"], rtol=0, atol=1e-12)
    assert gr["best_count"] > 0.6 * n and np.abs(gr["Rt"][:, 3] - [0.05, -0.02, -0.3]).max() < 0.02


SIZES = [  # name, mode, numDisparities, (w, h) crop or None, compute waves per strip, expected schedule
    ("T0", 0, 48, None, 7, "diag"),                 # padded disparity range (48 -> Dp 64)
    ("C1", 0, 64, None, 7, "diag"), ("C1", 1, 64, None, 7, "diag"), ("C1", 0, 64, None, 15, "diag"), ("C1", 1, 64, None, 15, "diag"),
    ("C1", 0, 112, None, 7, "diag"), ("C1", 1, 112, (632, 471), 15, "diag"),        # 112 -> Dp 128; a height that is no multiple of the unroll
    ("C1", 0, 96, None, 7, "diag"), ("C1", 0, 32, None, 15, "diag"),               # 3 and 1 registers per lane
    ("C1", 0, 160, None, 7, "diag"), ("C1", 1, 224, (632, 200), 7, "diag"),         # 5 and 7 registers per lane
    ("C1", 0, 64, (633, 471), 7, "ragged"), ("C1", 1, 64, (633, 471), 7, "ragged"),  # width - D = 569: W + E by k_sgbm_pair
    ("C1", 0, 64, (134, 59), 7, "ragged"), ("C1", 1, 112, (200, 59), 7, "diag"),     # lines shorter than two segments; fewer rows than a strip is wide
    ("C1", 0, 64, (79, 64), 15, "ragged"),                                        # width - D = 15 < one segment
    # 11 compute waves per strip: the width the look-ahead engines use by default (launch_diag)
    ("T0", 0, 48, None, 11, "diag"), ("C1", 0, 64, None, 11, "diag"), ("C1", 1, 64, None, 11, "diag"), ("C1", 1, 112, (632, 471), 11, "diag"),
    ("C1", 0, 32, None, 11, "diag"), ("C1", 0, 96, None, 11, "diag"), ("C1", 0, 64, (79, 64), 11, "ragged"), ("C1", 1, 112, (200, 59), 11, "diag"),
    # W + E with every row cut at its middle (k_sgbm_we2: width - D a multiple of 16): the smallest halves (two and three
    # segments), an odd number of four-row groups (the last workgroup's second group lies outside the image), a width - D that
    # is a multiple of 8 only (k_sgbm_we, whole rows) beside them
    ("C1", 0, 64, (96, 59), 7, "diag"), ("C1", 1, 64, (112, 41), 11, "diag"), ("C1", 0, 64, (104, 59), 7, "diag"), ("T0", 1, 32, (320, 237), 11, "diag"),
]


@pytest.mark.parametrize("name,mode,ndisp,crop,waves,sched", SIZES)
def test_diagonal_schedule_sizes_and_modes(oracle, name, mode, ndisp, crop, waves, sched, monkeypatch):
    """The aggregation schedule (W + E as one volume, then NW / N / NE + WTA in the diagonal sweep; MODE_HH with the
    reverse sweep first) against the oracle across register counts, padded disparity ranges, strip widths, ragged widths
    (k_sgbm_pair instead of k_sgbm_we / k_sgbm_we2) and images smaller than a strip."""
    c = Corridor(name)
    L, R = c.pair(4)
    if crop:
        L, R = np.ascontiguousarray(L[:crop[1], :crop[0]]), np.ascontiguousarray(R[:crop[1], :crop[0]])
    p = c.sgbm_params(mode)
    p["numDisparities"] = ndisp
    monkeypatch.setenv("VO_DIAG_WAVES", str(waves))
    h, w = L.shape
    ctx = _native.Context(0, max(w, 64), max(h, 64), max(16, ndisp), 64)
    ctx.set_sgbm(p, mode)
    got = ctx.sgbm_compute_host(L, R)
    again = ctx.sgbm_compute_host(L, R)                           # the boundary granules of the first run must not leak into the second
    status, schedule = ctx.sgbm_sweep_status(), ctx.sgbm_last_schedule()
    ctx.close()
    ref = oracle.sgbm_compute(L, R, p, mode)
    assert status == 0
    assert schedule == (_native.SCHED_DIAG if sched == "diag" else _native.SCHED_DIAG_RAGGED)
    assert np.array_equal(got, ref), int((got != ref).sum())
    assert np.array_equal(again, ref)


@pytest.mark.parametrize("mode", [0, 1])
def test_uniqueness_ratio_100_takes_the_unfused_schedule(oracle, mode):
    """uniquenessRatio >= 100 has no threshold form: one stored volume per direction (k_sgbm_paths) and the per-pixel
    winner search (k_sgbm_wta) -- against the oracle, and back to the fused schedule on the same context."""
    c = Corridor("T0")
    L, R = c.pair(3)
    p = c.sgbm_params(mode)
    p["uniquenessRatio"] = 100
    ctx = _native.Context(0, c.w, c.h, c.D, 64)
    ctx.set_sgbm(p, mode)
    got = ctx.sgbm_compute_host(L, R)
    assert ctx.sgbm_last_schedule() == _native.SCHED_UNFUSED
    assert np.array_equal(got, oracle.sgbm_compute(L, R, p, mode))
    p["uniquenessRatio"] = 99                                     # 100 - ur = 1: the reciprocal's special case
    ctx.set_sgbm(p, mode)
    got = ctx.sgbm_compute_host(L, R)
    assert ctx.sgbm_last_schedule() == _native.SCHED_DIAG
    assert np.array_equal(got, oracle.sgbm_compute(L, R, p, mode))
    ctx.close()


def test_c2_stream_of_24_pairs_default_policy_every_pair_against_the_oracle():
    """What the headline bench runs: a staged C2 stream through StereoOdometer with the default look-ahead (pairs computed
    ahead on the engines, each with the diagonal schedule) -- EVERY pair's disparity, keypoints and descriptors against
    the oracle, the chained pose to 1e-9."""
    from oracle.odometer import RefStereoCamera, RefStereoOdometer
    c = Corridor("C2")
    cam = StereoCamera(c.K(), c.dist(), c.K(), c.dist(), c.rect_params(), c.sgbm_params(), (c.w, c.h), max_keypoints=500)
    rcam = RefStereoCamera(cam.Q, cam.valid_region_left, c.sgbm_params())
    kw = dict(preprocessed_frames=True, rigidity_threshold=0.1, outlier_threshold=0.02)
    odo, rodo = StereoOdometer(cam, **kw), RefStereoOdometer(rcam, **kw)
    frames = c.pairs(30, 24)
    staged = cam.stage_pairs(frames)
    depth_seen = 0
    for k, (L, R) in enumerate(frames):
        a, b = odo.update(staged[k], None), rodo.update(L, R)
        depth_seen = max(depth_seen, cam._ctx.lookahead_depth())
        assert a == b and odo.skip_cause == rodo.skip_cause, k
        if a:
            _check_frame(odo, rodo, rcam, cam, k)
        assert np.allclose(odo.c_T_w, rodo.c_T_w, rtol=0, atol=1e-9), k
    assert depth_seen >= 4                                        # pairs really were computed ahead, behind a queue
    assert cam._ctx.sgbm_last_schedule() == _native.SCHED_DIAG and cam._ctx.sgbm_sweep_status() == 0


def test_c5_mono_pair_device_chain_and_odometer(oracle, c5):
    """BASELINE config 5 as ONE device-resident step (vo_mono_pair): slot keypoints -> kNN-2 -> ratio -> 5000-hypothesis
    essential RANSAC with a single host synchronisation; equals the stage-by-stage oracle composition bit for bit
    (M, winner, inlier count, mask, match indices; E to 1e-12).  Then MonoOdometer on four frames: the recovered
    rotation and translation direction follow the synthetic ground truth (no openVO oracle exists for this class)."""
    from openvo_amd.mono import MonoOdometer
    c, ctx, frames = c5
    for s, f in enumerate(frames):
        ctx.upload_mono(s, f)
        assert ctx.orb_slot_count(s, 8000, 0) > 7000
    K4 = [c.f, c.f, c.cx, c.cy]
    ref = [oracle.orb_detect_and_compute(f, None, 8000) for f in frames]
    ri, rd = oracle.bf_knn2_hamming(ref[0]["desc"], ref[1]["desc"])
    rq, rt = oracle.ratio_filter(ri, rd, 0.8)
    for solver in (8, 5):
        got = ctx.mono_pair(0, 1, 0.8, K4, 5000, 1.0, 4321, want_matches=True, solver=solver)
        rr = oracle.ransac_essential(ref[0]["xy"][rq], ref[1]["xy"][rt], K4, 5000, 1.0, 4321, solver=solver)
        assert got["matches"] == len(rq) and np.array_equal(got["q"], rq) and np.array_equal(got["t"], rt)
        assert got["best_iter"] == rr["best_iter"] and got["best_count"] == rr["best_count"], solver
        assert np.array_equal(got["mask"], rr["mask"]) and np.allclose(got["E"], rr["E"], rtol=0, atol=1e-12)
    K = np.array([[c.f, 0, c.cx], [0, c.f, c.cy], [0, 0, 1.0]])
    odo = MonoOdometer(K, (c.w, c.h), nfeatures=8000, context=ctx)            # five-point solver by default
    assert odo.solver == 5
    for k in range(4):
        assert odo.update(c.pair(k)[0]), (k, odo.skip_cause)
        if k:
            gt = np.linalg.inv(Corridor.gt_pose(k)) @ Corridor.gt_pose(k - 1)      # frame k-1 -> k, as c_T_w chains it
            Tk = odo.c_T_w @ np.linalg.inv(prev_c)
            dirg = gt[:3, 3] / np.linalg.norm(gt[:3, 3])
            assert np.dot(Tk[:3, 3], dirg) > 0.98                                     # translation direction within ~11 degrees
            assert np.abs(Tk[:3, :3] - gt[:3, :3]).max() < 0.02
            assert odo.last["best_count"] > 0.5 * odo.last["matches"]
        prev_c = odo.c_T_w.copy()


def test_stereo_pose_step_beyond_the_lds_resident_match_count():
    """More keypoints than the pose step's LDS-resident sets hold (> 3584: the four per-match arrays of the greedy
    clique move to the global workspace, the bit rows are read from HBM): two C2 frames with nfeatures = 6000 and the
    rigidity filter on, against the oracle odometer."""
    from oracle.odometer import RefStereoCamera, RefStereoOdometer
    c = Corridor("C2")
    cam = StereoCamera(c.K(), c.dist(), c.K(), c.dist(), c.rect_params(), c.sgbm_params(), (c.w, c.h), max_keypoints=6000)
    rcam = _CachedRefCamera(RefStereoCamera(cam.Q, cam.valid_region_left, c.sgbm_params()))
    kw = dict(nfeatures=6000, rigidity_threshold=0.1, outlier_threshold=0.02, preprocessed_frames=True)
    odo, rodo = StereoOdometer(cam, **kw), RefStereoOdometer(rcam, **kw)
    for k, (L, R) in enumerate(c.pairs(30, 2)):
        a, b = odo.update(L, R), rodo.update(L, R)
        assert a == b and odo.skip_cause == rodo.skip_cause, k
        _check_frame(odo, rodo, rcam, cam, k)
        assert np.allclose(odo.c_T_w, rodo.c_T_w, rtol=0, atol=1e-9), k
    assert len(odo.current_kps.xy) > 3584, len(odo.current_kps.xy)       # the case this test is about


def test_mono_lookahead_gives_the_same_chain():
    """MonoOdometer on a staged stream: the ORB extraction of the next frames runs ahead on look-ahead engines
    (vo_prefetch_staged_mono) while the current pair is matched and scored; accept flags, inlier counts and poses are
    identical to the one-frame-at-a-time run, also when a frame is asked for out of order."""
    from openvo_amd.mono import MonoOdometer
    c = Corridor("C5")
    K = np.array([[c.f, 0, c.cx], [0, c.f, c.cy], [0, 0, 1.0]])
    frames = [c.pair(k)[0] for k in range(7)]
    order = [0, 1, 2, 3, 5, 4, 6]                                      # one request the look-ahead did not predict
    chains = {}
    for depth in (0, 3):
        odo = MonoOdometer(K, (c.w, c.h), nfeatures=3000, ransac_iters=1500)
        odo.lookahead = depth
        odo.stage_frames(frames)
        chain = []
        for k in order:
            ok = odo.update(k)
            chain.append((ok, odo.skip_cause, None if odo.last is None else (odo.last["matches"], odo.last["best_iter"], odo.last["best_count"]), odo.c_T_w.copy()))
        chains[depth] = chain
        odo._ctx.close()
    for (a, ca, la, Ta), (b, cb, lb, Tb) in zip(chains[0], chains[3]):
        assert a == b and ca == cb and la == lb and np.array_equal(Ta, Tb)
    assert sum(x[0] for x in chains[3]) >= 6


def test_mono_pair_begin_end_equals_the_synchronous_step_and_tickets_end_in_any_order():
    """vo_mono_pair_begin / _end: three pairs in flight on alternates of their own give, bit for bit, what vo_mono_pair gives one at
    a time; tickets may be collected in any order; a fourth begin is refused while all are open; a slot read by an open
    ticket may be refilled (the refill is ordered behind the ticket on the device)."""
    c = Corridor("C5")
    K4 = [c.f, c.f, c.cx, c.cy]
    ctx = _native.Context(0, c.w, c.h, 16, 3000)
    frames = [c.pair(k)[0] for k in range(5)]
    for s, f in enumerate(frames[:4]):
        ctx.upload_mono(s, f)
        assert ctx.orb_slot_count(s, 3000, 0) > 1000
    want = [ctx.mono_pair(a, a + 1, 0.8, K4, 1500, 1.0, 4321, want_matches=True, solver=5) for a in range(3)]
    xy = [ctx.download_keypoints_xy(s) for s in range(4)]
    tickets = [ctx.mono_pair_begin(a, a + 1, 0.8, K4, 1500, 1.0, 4321, want_matches=True, solver=5) for a in range(3)]
    assert sorted(tickets) == [0, 1, 2]
    extra = [ctx.mono_pair_begin(0, 1, 0.8, K4, 1500, 1.0, 4321, want_matches=False, solver=5) for _ in range(_native.VO_NUM_MONO_ASYNC - 3)]
    with pytest.raises(Exception):                                     # every alternate is open
        ctx.mono_pair_begin(0, 1, 0.8, K4, 1500, 1.0, 4321, want_matches=True, solver=5)
    for t in extra:
        assert ctx.mono_pair_end(t)["best_count"] == want[0]["best_count"]
    ctx.upload_mono(0, frames[4])                                      # slot 0 is still being read by the first ticket
    for a in (1, 2, 0):
        got = ctx.mono_pair_end(tickets[a], want_matches=True)
        w = want[a]
        assert (got["matches"], got["best_iter"], got["best_count"]) == (w["matches"], w["best_iter"], w["best_count"]), a
        assert np.array_equal(got["E"], w["E"]) and np.array_equal(got["mask"], w["mask"]), a
        assert np.array_equal(got["q"], w["q"]) and np.array_equal(got["t"], w["t"]), a
        assert np.array_equal(got["xy_b"][:len(xy[a + 1])], xy[a + 1]), a
    with pytest.raises(Exception):
        ctx.mono_pair_end(tickets[0], want_matches=True)               # not open any more
    t = ctx.mono_pair_begin(1, 2, 0.8, K4, 1500, 1.0, 4321, want_matches=False, solver=8)   # alternates are free again
    got = ctx.mono_pair_end(t)
    ref = ctx.mono_pair(1, 2, 0.8, K4, 1500, 1.0, 4321, solver=8)
    assert np.array_equal(got["E"], ref["E"]) and got["best_count"] == ref["best_count"]
    ctx.close()


def test_mono_speculative_pair_steps_survive_a_rejected_frame():
    """MonoOdometer begins the steps of the next pairs before it has collected the current one's, assuming every frame will
    be accepted.  A frame that is rejected (a blank image: no keypoints; an image of noise: too few matches / inliers) voids what
    was begun ahead; flags, counts and poses equal the run without speculation, and both outcomes really occurred."""
    from openvo_amd.mono import MonoOdometer
    c = Corridor("C5")
    K = np.array([[c.f, 0, c.cx], [0, c.f, c.cy], [0, 0, 1.0]])
    frames = [c.pair(k)[0] for k in range(10)]
    frames[3] = np.zeros_like(frames[3])                                # no keypoints
    frames[6] = np.random.default_rng(5).integers(0, 256, frames[6].shape, dtype=np.uint8)   # noise: keypoints, but no consistent matches
    chains, stats = {}, {}
    for spec in (0, 2):
        odo = MonoOdometer(K, (c.w, c.h), nfeatures=3000, ransac_iters=1500, min_inliers=200)
        odo.speculate = spec
        odo.stage_frames(frames)
        chain = []
        for k in range(len(frames)):
            ok = odo.update(k)
            chain.append((ok, odo.skip_cause if not ok else "", None if odo.last is None else (odo.last["matches"], odo.last["best_iter"], odo.last["best_count"]), odo.c_T_w.copy()))
        chains[spec], stats[spec] = chain, dict(odo.speculation)
        odo.close()
    for (a, ca, la, Ta), (b, cb, lb, Tb) in zip(chains[0], chains[2]):
        assert a == b and ca == cb and la == lb and np.array_equal(Ta, Tb)
    flags = [x[0] for x in chains[2]]
    assert flags[3] is False and chains[2][3][1] == "keypoints"
    assert flags[6] is False and chains[2][6][1] in ("inliers", "matches")
    assert sum(flags) >= 7
    assert stats[0] == {"begun": 0, "used": 0, "void": 0}
    assert stats[2]["used"] >= 3 and stats[2]["void"] >= 1, stats[2]


def test_mono_unpredicted_request_while_predictions_are_in_flight():
    """update(0) starts the extraction of frames 1..3 on look-ahead engines; update(5) -- a frame nobody predicted -- voids
    them while they may still be running and immediately reuses one of their slots on another stream.  The slot's next use
    must be ordered behind the voided extraction (vo_prefetch_staged_mono / the upload wait for the slot's `ready` event):
    the chain equals the one-frame-at-a-time run."""
    from openvo_amd.mono import MonoOdometer
    c = Corridor("C5")
    K = np.array([[c.f, 0, c.cx], [0, c.f, c.cy], [0, 0, 1.0]])
    frames = [c.pair(k)[0] for k in range(8)]
    order = [0, 5, 6, 2, 7, 3]                                         # every second request is one the look-ahead did not predict
    chains = {}
    for depth in (0, 3):
        odo = MonoOdometer(K, (c.w, c.h), nfeatures=3000, ransac_iters=1500)
        odo.lookahead = depth
        odo.stage_frames(frames)
        chain = []
        for k in order:
            ok = odo.update(k)
            n = odo._ctx.orb_slot_count(odo._ref[0], odo.nfeatures, 0)
            chain.append((ok, odo.skip_cause, n, None if odo.last is None else (odo.last["matches"], odo.last["best_count"]), odo.c_T_w.copy()))
        chains[depth] = chain
        assert odo._ctx.lookahead_depth() <= depth
        odo._ctx.close()
    for (a, ca, na, la, Ta), (b, cb, nb, lb, Tb) in zip(chains[0], chains[3]):
        assert a == b and ca == cb and na == nb and la == lb and np.array_equal(Ta, Tb)


@pytest.mark.parametrize("mode", [0, 1])
def test_sgbm_4k_matches_the_oracle_fixture(mode):
    """Beyond every BASELINE size: 3840 x 2160 with D = 256 -- a 3.96 GB cost volume, so that every 32-bit byte offset inside the
    volume kernels (buffer descriptors, scalar offsets) is exercised past 2^31 -- in MODE_SGBM and in MODE_HH (the reverse sweep
    writes a second volume).  The oracle needs minutes and tens of GB for this frame, so its result travels as a fixture
    (tests/golden/make_oracle_4k.py: SHA-256 of the whole disparity image + eight of its rows per mode; an ORACLE-derived
    regression fixture, not a reference-derived one); the pair itself is rebuilt from integer arithmetic (tests/big_pair.py).
    Also: the same call twice gives the same bytes, the ground plane's disparity law holds, and a size whose volume would pass
    4 GiB is refused, not truncated."""
    import hashlib, os
    from tests.big_pair import pair
    fx = np.load(os.path.join(os.path.dirname(__file__), "golden", "o1_sgbm_4k.npz"))
    w, h = int(fx["w"]), int(fx["h"])
    p = {str(k): int(v) for k, v in zip(fx["param_names"], fx["params"])}
    L, R, d = pair(w, h)
    ctx = _native.Context(0, 4096, 2304, p["numDisparities"], 500)
    try:
        ctx.set_sgbm(p, mode)
        got = ctx.sgbm_compute_host(L, R)
        again = ctx.sgbm_compute_host(L, R)
        if mode == 0:
            big = np.zeros((2304, 4096), np.uint8)
            with pytest.raises(_native.VoError, match="4 GiB"):
                ctx.sgbm_compute_host(big, big)
    finally:
        ctx.close()
    assert got.shape == (h, w) and np.array_equal(got, again)
    rows = [int(r) for r in fx["rows"]]
    for r, want in zip(rows, fx["disp_rows_mode%d" % mode]):
        assert np.array_equal(got[r], want), "row %d: %d pixels differ" % (r, int((got[r] != want).sum()))
    assert hashlib.sha256(got.tobytes()).digest() == fx["sha256_mode%d" % mode].tobytes()
    mid = got[h // 2][got[h // 2] >= 0] / 16.0
    assert abs(np.median(mid) - d[h // 2]) < 0.5


@pytest.mark.parametrize("nfeatures", [2000, 8000])
def test_orb_4k_matches_the_oracle_fixture(nfeatures):
    """ORB on a 3840 x 2160 image (the pyramid's cones, the FAST tiles and the candidate lists at four times config 5's pixel
    count), through the fused selection kernel (nfeatures <= 2000) and through the three-launch one: keypoints, responses,
    angles, octaves and descriptors bit for bit the oracle's -- whose run on this image takes 100 s per call, so it travels as
    digests of every output array (tests/golden/make_oracle_4k.py orb; oracle-derived, not reference-derived)."""
    import hashlib, os
    from tests.big_pair import pair
    fx = np.load(os.path.join(os.path.dirname(__file__), "golden", "o2_orb_4k.npz"))
    L, _, _ = pair(int(fx["w"]), int(fx["h"]), seed=int(fx["seed"]))
    ctx = _native.Context(0, 3840, 2160, 64, nfeatures)
    try:
        got = ctx.orb_host(L, None, nfeatures)
    finally:
        ctx.close()
    n = nfeatures
    assert len(got["xy"]) == int(fx["count_%d" % n]) >= 0.9 * n
    assert np.array_equal(np.bincount(got["octave"], minlength=8), fx["per_level_%d" % n])
    assert np.array_equal(got["xy"][:64].view(np.uint32), fx["head_xy_%d" % n].view(np.uint32))
    assert np.array_equal(got["desc"][:64], fx["head_desc_%d" % n])
    DT = dict(xy=np.float32, response=np.float32, angle=np.float32, octave=np.int32, desc=np.uint8)
    for k in ("xy", "response", "angle", "octave", "desc"):
        assert hashlib.sha256(np.ascontiguousarray(got[k], dtype=DT[k]).tobytes()).digest() == fx["sha256_%s_%d" % (k, n)].tobytes(), k
