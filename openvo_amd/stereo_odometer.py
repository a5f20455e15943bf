"""StereoOdometer: drop-in for openVO's class (reference stereo_odometer.py:4-226).

Same constructor defaults, class constants, attributes, methods, return values and
`skip_cause` strings.  The per-frame arithmetic (disparity mask + ORB, Hamming kNN + ratio
test, 3-D lookup, rigid-body clique filter, Umeyama fit) runs in libvo355 on the GPU; what
stays in Python is the sequential bookkeeping of `update` -- it decides which frames pair up,
so it follows the reference decision for decision (pinned by tests/golden/g5_state_machine.json).
"""
import os

import numpy as np

from . import _native
from .features import BFMatcher, DeviceImage, DisparityMask, KeyPointList, ORB


class StereoOdometer:
    # disparity range (pixels) with a usable depth estimate          [reference :6-7]
    MIN_VALID_DISPARITY = 4
    MAX_VALID_DISPARITY = 100
    # per-frame motion gates, widened by (skipped_frames + 1)          [reference :10,12]
    MAX_DISTANCE_CHANGE = 1          # metres
    MAX_ROTATION_CHANGE = np.pi / 3  # radians

    def __init__(self, stereo_camera, nfeatures=500, match_threshold=0.8, rigidity_threshold=0,
                 outlier_threshold=0, preprocessed_frames=False, min_matches=10,
                 pose_method="umeyama", pnp_iters=256, pnp_threshold=1.5, pnp_seed=4321):
        """Arguments up to min_matches are the reference's [reference :14-15].  pose_method="pnp" is an
        extension (not in openVO): the pair's pose comes from RANSAC solvePnP on the previous frame's 3-D
        points and the new frame's keypoint pixels (vo_ransac_pnp) instead of the 3-D/3-D Umeyama fit;
        the rigidity / outlier stages are then not used, the motion gates still apply."""
        if pose_method not in ("umeyama", "pnp"):
            raise ValueError("pose_method must be 'umeyama' or 'pnp'")
        self.pose_method, self.pnp_iters, self.pnp_threshold, self.pnp_seed = pose_method, pnp_iters, pnp_threshold, pnp_seed
        self.stereo = stereo_camera
        self.current_img = self.current_disparity = self.current_3d = None
        self.prev_img = self.prev_disparity = self.prev_3d = None
        ctx = getattr(stereo_camera, "_ctx", None)
        self._ctx = ctx
        self.orb = ORB(ctx, nfeatures) if ctx is not None else None
        self.matcher = BFMatcher(ctx) if ctx is not None else None
        self.prev_kps = self.current_kps = None
        self.current_desc = None
        self.match_threshold = match_threshold
        self.rigidity_threshold = rigidity_threshold
        self.outlier_threshold = outlier_threshold
        self.preprocessed_frames = preprocessed_frames
        self.min_matches = min_matches
        self.skipped_frames = 0      # successive frames without an accepted transform
        self.c_T_w = np.eye(4)       # world frame expressed in the camera frame
        self.c_T_w_prev = np.eye(4)
        self.skip_cause = ""
        self._specs = {}             # (slot key a, slot key b, params) -> ticket of a pose step started ahead of time
        self._next_hint = ()         # SubmittedPairs expected by the next update() calls (set by run())
        self._ahead_counts = {}      # (slot, generation), ORB arguments -> keypoint count of a look-ahead pair already collected
        self.pose_ahead = max(0, min(int(os.environ.get("VO_POSE_AHEAD", "4")), _native.VO_NUM_POSE_ASYNC - 1))   # 4 against 6 / 5 / 3 / 2: DESIGN 4b

    # ------------------------------------------------------------------------------------------
    def feature_mask(self, disparity):
        """uint8 {0,255} mask of MIN_VALID_DISPARITY <= d <= MAX_VALID_DISPARITY  [reference :38-41].
        For a device-resident disparity the mask is returned unevaluated and fused into ORB."""
        if isinstance(disparity, DeviceImage) and disparity.kind == "disp" and disparity.frame.live:
            return DisparityMask(disparity.frame, self.MIN_VALID_DISPARITY, self.MAX_VALID_DISPARITY)
        d = np.asarray(disparity)
        return ((d >= self.MIN_VALID_DISPARITY) * (d <= self.MAX_VALID_DISPARITY)).astype(np.uint8) * 255

    def valid_distance_change(self, prev_kp_idx, current_kp_idx):
        """Unused by the reference (guarded by `if (False)`, :165-166); kept for API parity [:43-48]."""
        p_x, p_y = self.prev_kps[prev_kp_idx].pt
        c_x, c_y = self.current_kps[current_kp_idx].pt
        a = np.linalg.norm(np.asarray(self.prev_3d)[int(p_y)][int(p_x)])
        b = np.linalg.norm(np.asarray(self.current_3d)[int(c_y)][int(c_x)])
        return a - b <= self.MAX_DISTANCE_CHANGE * (self.skipped_frames + 1)

    def bilinear_interpolate_pixels(self, img, x, y):
        """Inf-aware bilinear sample of an HxWx3 image at float (x, y)  [reference :50-79].
        Raises ZeroDivisionError when no tap is usable, like the reference."""
        if isinstance(img, DeviceImage) and img.kind == "xyz" and img.frame.live:
            out, st = self._ctx.points3d_at(img.frame.slot, np.array([[x, y]], np.float32))
        else:
            out, st = self._ctx.bilinear_at(np.asarray(img, np.float32), np.array([[x, y]], np.float32))
        if st[0] == 2:
            raise ZeroDivisionError("division by zero")
        return out[0]

    def rigid_body_filter(self, prev_pts, pts):
        """Greedy maximum clique of pairwise-distance-consistent matches  [reference :82-105]."""
        return self._ctx.rigid_clique(np.asarray(prev_pts, np.float32), np.asarray(pts, np.float32),
                                      self.rigidity_threshold)

    def save_frame_update(self, next_img, next_disp, next_3d, next_kps, next_desc):
        self.prev_img, self.prev_disparity, self.prev_3d = self.current_img, self.current_disparity, self.current_3d
        self.prev_kps, self.prev_desc = self.current_kps, self.current_desc
        self.current_img, self.current_disparity, self.current_3d = next_img, next_disp, next_3d
        self.current_kps, self.current_desc = next_kps, next_desc

    # ------------------------------------------------------------------------------------------
    def update(self, img_left, img_right):
        """Process one stereo pair; True when the frame was accepted  [reference :115-160].

        Raises _native.SweepTimeout (a VoError) instead of returning when the pair's disparity is undefined -- a strip
        hand-off inside its aggregation sweep gave up waiting on an oversubscribed GPU: no pose is ever derived from such a
        pair.  The odometer's state is that of before the call (the frame is neither saved nor counted as skipped); the same
        pair may be passed again, or the next one."""
        next_3d, next_disp, next_img = self.stereo.compute_3d(img_left, img_right,
                                                              preprocessed=self.preprocessed_frames)
        next_kps, next_desc = self.orb.detectAndCompute(next_img, self.feature_mask(next_disp))
        if len(next_kps) < self.min_matches:
            self.skipped_frames += 1
            self.skip_cause = "keypoints"
            return False
        if self.current_img is None:           # very first usable frame
            self.save_frame_update(next_img, next_disp, next_3d, next_kps, next_desc)
            return True

        T = self._try_pair(self.current_kps, self.current_desc, self.current_3d, next_kps, next_desc, next_3d)
        if T is not None:
            self.c_T_w_prev = self.c_T_w
            self.c_T_w = T @ self.c_T_w
        elif self.prev_img is not None:
            # one-frame-back fallback: pair the new frame with the frame before `current`
            T = self._try_pair(self.prev_kps, self.prev_desc, self.prev_3d, next_kps, next_desc, next_3d)
            if T is not None:
                base = self.c_T_w_prev
                self.c_T_w_prev = self.c_T_w
                self.c_T_w = T @ base
                self.skipped_frames = 0
        if T is None:
            self.skipped_frames += 1          # frame is dropped, `current` stays
            return False
        self.skipped_frames = 0
        self.save_frame_update(next_img, next_disp, next_3d, next_kps, next_desc)
        self._start_next_pose()
        return True

    def _pose_params(self):
        return (float(self.match_threshold), int(self.min_matches), float(max(self.rigidity_threshold, 0)),
                float(max(self.outlier_threshold, 0)))

    def _fused_ok(self):
        return (type(self) is StereoOdometer and type(self.matcher) is BFMatcher and self.pose_method == "umeyama"
                and not any(n in self.__dict__ for n in self._SEAMS))

    def _drop_specs(self, keep=()):
        """Collect and discard pose steps begun ahead.  A step's own error (e.g. VO_E_SWEEP on a pair it read) is discarded with
        it: it belongs to a result nobody asked for, and that pair's own update() reports what is wrong with it."""
        for key in [k for k in self._specs if k not in keep]:
            try:
                self._ctx.pose_pair_end(self._specs[key])
            except _native.VoError:
                pass
            finally:
                del self._specs[key]

    def reset_lookahead(self):
        """Collect and drop the pose steps begun ahead of time and the camera's unconsumed look-ahead pairs (they are
        recomputed when asked for).  Not in the reference: lets a caller start a measurement, or hand the camera to another
        odometer, with nothing in flight."""
        if self._ctx is not None:
            self._drop_specs()
        if hasattr(self.stereo, "reset_lookahead"):
            self.stereo.reset_lookahead()

    def _start_next_pose(self):
        """The frame just accepted is the new `current`.  The pairs that will come next may already be on the device
        (look-ahead / submitted ahead): for as many of them as have FINISHED their disparity and keypoints (never waiting for
        one), start the matching + pose step now on a stream of its own -- (current, next), and, expecting every frame to be
        accepted, (next, next+1), ... up to `pose_ahead` steps.  In the steady state that is the next pair or none; when
        results arrive in a burst (a cold start: every pair in flight finishes at about the same time) the short kernel
        chains of many pairs run side by side instead of one after the other behind the host.  Purely an ordering change:
        _pair_fused looks a step up by its slots and parameters and computes on the spot when the guess was wrong."""
        if not self._fused_ok() or self.orb.last_slot_args is None:
            return self._drop_specs()
        kps = self.current_kps
        if not self._on_device(kps, self.current_desc, self.current_3d):
            return self._drop_specs()
        depth = self.pose_ahead
        nxt = [h.slot for h in self._next_hint] if self._next_hint else self.stereo.next_lookahead_slots(depth)
        chain, counts = [kps.frame.slot], [len(kps)]
        for s in nxt[:depth]:
            if s is None or not self._ctx.slot_ready(s):
                break
            chain.append(s)
            key = (self.stereo.slot_key(s), self.orb.last_slot_args)
            if key not in self._ahead_counts:
                if len(self._ahead_counts) > 64:
                    self._ahead_counts.clear()
                try:
                    self._ahead_counts[key] = self._ctx.orb_slot_count(s, *self.orb.last_slot_args)   # (finished: only collects the count)
                except _native.VoError:
                    self._ahead_counts[key] = -1     # that pair's own update() reports what is wrong with it; nothing is begun on it here
            counts.append(self._ahead_counts[key])
        params = self._pose_params()
        sk = self.stereo.slot_key       # slot + generation: a slot refilled with another pair never matches
        wanted = []
        for j in range(len(chain) - 1):
            if not (0 < counts[j] <= 3800 and counts[j + 1] >= max(2, self.min_matches)):
                break                    # (a frame that will be rejected: what lies behind it pairs with another reference)
            wanted.append((sk(chain[j]), sk(chain[j + 1]), params))
        self._drop_specs(keep=wanted)
        from ._native import VoError
        for key in wanted:
            if key not in self._specs and len(self._specs) < _native.VO_NUM_POSE_ASYNC:
                try:
                    self._specs[key] = self._ctx.pose_pair_begin(key[0][0], key[1][0], *params)
                except VoError:
                    break                    # nothing started: update() computes the step when it gets there

    def run(self, pairs, depth=None, on_sweep_timeout="raise"):
        """Feed an iterable of host (left, right) pairs through update(), keeping up to `depth` pairs
        submitted ahead so their upload and disparity overlap the tracking of the current pair.  Yields
        update()'s result per pair, in order.  Not in the reference (whose update() takes one host pair per call,
        stereo_odometer.py:115-116).

        The copy of each pair into pinned staging memory runs on the library's own staging thread (vo_host_stage_begin), one
        pair ahead of the pair being submitted: it overlaps this thread's kernel launches and waits, this thread never touches
        image bytes, and no second Python thread competes for the interpreter lock.  The caller may refill the arrays it
        yielded as soon as it is asked for the next pair (a copy is waited for before the iterator is advanced).  Pairs whose
        two images differ in channel count go through StereoCamera.submit() instead.

        A pair whose disparity is undefined (update() raises SweepTimeout, see there): with on_sweep_timeout="raise" (default)
        the exception leaves the generator, which ends -- the failed pair and the pairs already taken from `pairs` but not yet
        yielded (at most depth + 2) are lost to this run, every frame slot they held is given back, the odometer's state is
        that of the last yielded result, and a new run() may follow at once; with "skip" the generator yields None for that
        pair (neither accepted nor counted as skipped: the odometer's state is untouched) and carries on with the next one.
        Either way no slot stays reserved when the generator ends, however it ends (exhausted, closed, or by an exception)."""
        from collections import deque
        if on_sweep_timeout not in ("raise", "skip"):
            raise ValueError("on_sweep_timeout must be 'raise' or 'skip'")
        cam, ctx = self.stereo, self.stereo._ctx
        depth = int(cam.lookahead if depth is None else depth)
        nbuf = _native.VO_NUM_HOST_STAGE
        it = iter(pairs)
        copying, queue = deque(), deque()            # staged (or being staged) and not yet submitted / submitted and not yet consumed
        state = {"k": 0, "done": False, "inflight": None}

        def stage_next():
            if state["inflight"] is not None:
                ctx.host_stage_wait(state["inflight"])                   # its source arrays are the caller's again
                state["inflight"] = None
            try:
                L, R = next(it)
            except StopIteration:
                state["done"] = True
                return
            L, R = np.asarray(L), np.asarray(R)
            if L.ndim != R.ndim or L.shape != R.shape:
                copying.append((None, L.copy(), R.copy()))               # mixed inputs: converted when submitted
            else:
                buf = state["k"] % nbuf
                copying.append((buf,) + ctx.host_stage_begin(buf, L, R))
                state["inflight"] = buf
                state["k"] += 1

        def keep_one_ahead():
            while not state["done"] and len(copying) < 2:
                stage_next()

        try:
            while True:
                keep_one_ahead()
                while len(queue) <= depth and copying and (len(copying) >= 2 or state["done"]):
                    item = copying.popleft()
                    if item[0] is None:
                        queue.append(cam.submit(item[1], item[2], preprocessed=self.preprocessed_frames))
                    else:
                        buf, w, h, ch, _keep = item
                        if state["inflight"] == buf:
                            state["inflight"] = None                     # (submit_staged waits for the copy)
                        queue.append(cam.submit_staged(buf, w, h, ch, self.preprocessed_frames))
                    keep_one_ahead()
                if not queue:
                    return
                head = queue.popleft()
                self._next_hint = tuple(queue)[:self.pose_ahead]
                try:
                    try:
                        res = self.update(head, None)
                    except _native.SweepTimeout:
                        if on_sweep_timeout != "skip":
                            raise
                        res = None
                    yield res
                finally:
                    self._next_hint = ()
        finally:
            if state["inflight"] is not None:                            # a copy begun and never submitted: its sources may go away now
                ctx.host_stage_wait(state["inflight"])
            # pairs submitted ahead that nobody will consume (the generator was closed early, or an exception ended it): their
            # slots are marked reserved and only a consumer ever un-marks one -- give them back, or up to depth + 1 of the 28
            # slots stay stranded and later submissions fall back to synchronous processing.  Pose steps begun ahead on them
            # are collected first (they read the slots); whatever still runs is ordered before a slot's next use.
            if any(sp.slot is not None for sp in queue):
                try:
                    self._drop_specs()
                finally:
                    for sp in queue:
                        cam.release_submitted(sp)
                    queue.clear()

    _SEAMS = ("point_clouds", "point_cloud_transform", "rigid_body_filter", "bilinear_interpolate_pixels",
              "_estimate", "_gate")

    def _try_pair(self, kps_a, desc_a, im3d_a, kps_b, desc_b, im3d_b):
        if self.pose_method == "pnp":
            return self._pair_pnp(kps_a, desc_a, im3d_a, kps_b, desc_b, im3d_b)
        # fused device path when nothing along the way was replaced by the user
        if (type(self) is StereoOdometer and type(self.matcher) is BFMatcher and 2 <= len(kps_b) and len(kps_a) <= 3800
                and self._on_device(kps_a, desc_a, im3d_a) and self._on_device(kps_b, desc_b, im3d_b)
                and not any(n in self.__dict__ for n in self._SEAMS)):
            return self._pair_fused(kps_a.frame.slot, kps_b.frame.slot)
        pts_a, pts_b = self.point_clouds(kps_a, kps_b, desc_a, desc_b, im3d_a, im3d_b)
        if pts_a is None:
            self.skip_cause = "matches"
            return None
        return self.point_cloud_transform(pts_a, pts_b)

    def _pair_pnp(self, kps_a, desc_a, im3d_a, kps_b, desc_b, im3d_b):
        """Extension: pose of the pair by RANSAC solvePnP (3-D of frame a, pixels of frame b)."""
        if not (self._on_device(kps_a, desc_a, im3d_a) and self._on_device(kps_b, desc_b, im3d_b) and len(kps_b) >= 2):
            raise ValueError("pose_method='pnp' needs the device-resident frames compute_3d returns")
        q, t, pts_a, _, st_a, _ = self._ctx.point_clouds(kps_a.frame.slot, kps_b.frame.slot, self.match_threshold)
        if len(q) < self.min_matches:
            self.skip_cause = "matches"
            return None
        if (st_a == 2).any():
            raise ZeroDivisionError("division by zero")
        ok = (st_a == 0) & np.isfinite(pts_a).all(axis=1)
        x0, y0 = kps_b.frame.roi[0], kps_b.frame.roi[1]
        uv = kps_b.xy[t[ok]] + np.array([x0, y0], np.float32)      # keypoints live in the cropped image
        Q = self.stereo.Q
        K4 = [Q[2, 3], Q[2, 3], -Q[0, 3], -Q[1, 3]]
        if ok.sum() < max(self.min_matches, 4):
            self.skip_cause = "matches"
            return None
        r = self._ctx.ransac_pnp(pts_a[ok], uv, K4, self.pnp_iters, self.pnp_threshold, self.pnp_seed)
        if r["best_count"] < self.min_matches:
            self.skip_cause = "outlier"
            return None
        return self._gate(np.vstack([r["Rt"], [0, 0, 0, 1]]))

    def _pair_fused(self, slot_a, slot_b):
        """point_clouds + point_cloud_transform in one native call (one device synchronisation);
        same decisions and skip_cause strings as the two methods."""
        from ._native import VoError
        params = self._pose_params()
        ticket = self._specs.pop((self.stereo.slot_key(slot_a), self.stereo.slot_key(slot_b), params), None)
        if ticket is not None:
            counts, rc, _, T34 = self._ctx.pose_pair_end(ticket)         # started by an earlier update()
        else:
            counts, rc, _, T34 = self._ctx.pose_pair(slot_a, slot_b, *params)
        M, n1, n2, flags = (int(v) for v in counts)
        if M < self.min_matches:
            self.skip_cause = "matches"
            return None
        if flags & 1:
            raise ZeroDivisionError("division by zero")
        too_few_rigid = n1 < 10
        if too_few_rigid:
            self.skip_cause = "rigidity"
        if rc[0] < 0:
            raise VoError(-5, "Points cannot be colinear" if rc[0] == -2 else "Umeyama needs at least 3 points")
        if n2 < self.min_matches:
            if not too_few_rigid:
                self.skip_cause = "outlier"
            return None
        if rc[1] < 0:
            raise VoError(-5, "Points cannot be colinear" if rc[1] == -2 else "Umeyama needs at least 3 points")
        return self._gate(np.vstack([T34, [0, 0, 0, 1]]))

    # ------------------------------------------------------------------------------------------
    def point_clouds(self, kps1, kps2, desc1, desc2, im3d1, im3d2):
        """Matched 3-D points of two frames (Mx3 float32 each) or (None, None)  [reference :162-175]."""
        fused = (type(self.matcher) is BFMatcher and self._on_device(kps1, desc1, im3d1)
                 and self._on_device(kps2, desc2, im3d2) and len(kps2) >= 2)
        if fused:
            q, t, pts1, pts2, st1, st2 = self._ctx.point_clouds(kps1.frame.slot, kps2.frame.slot,
                                                                self.match_threshold)
            if len(q) < self.min_matches:
                return None, None
            if (st1 == 2).any() or (st2 == 2).any():
                raise ZeroDivisionError("division by zero")
            return pts1, pts2
        # generic path: the same steps through the public seams (any matcher / arrays)
        matches = self.matcher.knnMatch(desc1, desc2, k=2)
        matches = [m[0] for m in matches if m[0].distance < self.match_threshold * m[1].distance]
        if len(matches) < self.min_matches:
            return None, None
        pts1 = [self.bilinear_interpolate_pixels(im3d1, *kps1[m.queryIdx].pt) for m in matches]
        pts2 = [self.bilinear_interpolate_pixels(im3d2, *kps2[m.trainIdx].pt) for m in matches]
        return np.array(pts1), np.array(pts2)

    @staticmethod
    def _on_device(kps, desc, im3d):
        return (isinstance(kps, KeyPointList) and kps.frame is not None and kps.frame.live
                and desc is kps.desc and isinstance(im3d, DeviceImage) and im3d.frame is kps.frame)

    def _estimate(self, src, dst):
        """cv2.estimateAffine3D(src, dst, force_rotation=True)[0] with the homogeneous row appended."""
        T34, _ = self._ctx.umeyama(src, dst, True)
        return np.vstack([T34, [0, 0, 0, 1]])

    def point_cloud_transform(self, current_pts, next_pts):
        """Rigid transform current -> next (4x4) or None with skip_cause set  [reference :177-223]."""
        if self.rigidity_threshold > 0:
            keep = self.rigid_body_filter(current_pts, next_pts) > 0
            current_pts, next_pts = current_pts[keep], next_pts[keep]
        too_few_rigid = len(current_pts) < 10
        if too_few_rigid:
            self.skip_cause = "rigidity"
        if self.outlier_threshold > 0 and not too_few_rigid:
            # single-pass outlier rejection on the relative residual of a first fit  [:188-197]
            T = self._estimate(current_pts, next_pts)
            h_next = np.hstack([next_pts, np.ones((len(next_pts), 1))]).astype(np.float64)
            h_cur = np.hstack([current_pts, np.ones((len(current_pts), 1))]).astype(np.float64)
            errors = np.linalg.norm(h_next - h_cur @ T.T, axis=1) / np.linalg.norm(h_next, axis=1)
            keep = errors < self.outlier_threshold + np.median(errors)
            current_pts, next_pts = current_pts[keep], next_pts[keep]
        if len(current_pts) < self.min_matches:
            if not too_few_rigid:
                self.skip_cause = "outlier"
            return None
        return self._gate(self._estimate(current_pts, next_pts))

    def _gate(self, T):
        """NaN check and the motion gates of the reference [:207-223]."""
        if np.isnan(T).any():
            self.skip_cause = "nan"
            return None
        scale = self.skipped_frames + 1
        dist = np.linalg.norm(T[0:3, 3])
        angle = np.linalg.norm(self._ctx.rodrigues(T[0:3, 0:3]))
        too_far = dist > self.MAX_DISTANCE_CHANGE * scale
        too_turned = angle > self.MAX_ROTATION_CHANGE * scale
        if too_far:
            self.skip_cause = "bigdist"
        if too_turned:
            self.skip_cause = "bigrot"
        if too_far or too_turned:
            return None
        return T

    def current_pose(self):
        """Camera pose in world (= first accepted frame) coordinates  [reference :225-226]."""
        return np.linalg.inv(self.c_T_w)
