"""CPU restatement of the reference's per-frame path, composed from the oracle's C functions.

TEST INFRASTRUCTURE ONLY (see oracle/vo_oracle.h): used by tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg as the checker / reported baseline.  Not imported by openvo_amd/.

Follows /root/reference/src/openVO/stereo_camera.py:43-55 (compute_3d) and
stereo_odometer.py:115-223 (update, point_clouds, point_cloud_transform) step by step, with
every cv2 call replaced by its oracle restatement.  Rectification setup is taken as given
(Q, roi, optional maps) -- it is one-off and outside the hot path.
"""
import numpy as np

from . import oracle as O


class RefStereoCamera:
    def __init__(self, Q, roi, sgbm_params, maps=None, mode=0, dense_3d=False):
        """dense_3d: also evaluate reprojectImageTo3D on the whole image, as the reference does on every
        pair (stereo_camera.py:52) -- the timed CPU baseline sets it; the sampled values are the same."""
        self.Q = np.asarray(Q, np.float64)
        self.valid_region_left = tuple(roi)
        self.sgbm_params, self.mode, self.maps = dict(sgbm_params), mode, maps
        self.dense_3d = dense_3d

    def crop(self, img):
        vr = self.valid_region_left
        return img[vr[1]: vr[3], vr[0]: vr[2]]

    def compute_3d(self, img_left, img_right, preprocessed=False):
        if img_left.ndim == 3:                                   # stereo_camera.py:44-47
            img_left = O.bgr2gray(img_left)
        if img_right.ndim == 3:
            img_right = O.bgr2gray(img_right)
        if not preprocessed:                                     # :48-50
            img_left = O.remap_bilinear(img_left, *self.maps[0])
            img_right = O.remap_bilinear(img_right, *self.maps[1])
        disp16 = O.sgbm_compute(img_left, img_right, self.sgbm_params, self.mode)
        disparity = disp16.astype(np.float32) / 16               # :51
        self.last_disp16 = disp16
        img_3d = _Lazy3D(disp16, self.Q, self.valid_region_left)  # :52 (evaluated where sampled)
        if self.dense_3d:
            with np.errstate(all="ignore"):
                self.last_dense = O.reproject_to_3d(disparity, self.Q)   # the full image, like cv2.reprojectImageTo3D
        return img_3d, self.crop(disparity), self.crop(img_left)


class _Lazy3D:
    """reprojectImageTo3D + crop, evaluated only at the sampled taps (identical values)."""

    def __init__(self, disp16, Q, roi):
        self.disp16, self.Q, self.roi = disp16, Q, roi

    def sample(self, xy):
        return O.points3d_at(self.disp16, self.Q, self.roi, xy)

    def dense(self):
        vr = self.roi
        full = O.reproject_to_3d(self.disp16.astype(np.float32) / 16, self.Q)
        return full[vr[1]: vr[3], vr[0]: vr[2]]


class RefStereoOdometer:
    MIN_VALID_DISPARITY, MAX_VALID_DISPARITY = 4, 100
    MAX_DISTANCE_CHANGE, MAX_ROTATION_CHANGE = 1, np.pi / 3

    def __init__(self, stereo_camera, nfeatures=500, match_threshold=0.8, rigidity_threshold=0,
                 outlier_threshold=0, preprocessed_frames=False, min_matches=10):
        self.stereo, self.nfeatures = stereo_camera, nfeatures
        self.match_threshold, self.rigidity_threshold = match_threshold, rigidity_threshold
        self.outlier_threshold, self.preprocessed_frames = outlier_threshold, preprocessed_frames
        self.min_matches = min_matches
        self.cur = self.prev = None          # dict(img, disp, d3, kps, desc)
        self.skipped_frames = 0
        self.c_T_w, self.c_T_w_prev = np.eye(4), np.eye(4)
        self.skip_cause = ""

    def feature_mask(self, disparity):                           # stereo_odometer.py:38-41
        m = (disparity >= self.MIN_VALID_DISPARITY) * (disparity <= self.MAX_VALID_DISPARITY)
        return m.astype(np.uint8) * 255

    def update(self, img_left, img_right):                       # :115-160
        d3, disp, img = self.stereo.compute_3d(img_left, img_right, preprocessed=self.preprocessed_frames)
        k = O.orb_detect_and_compute(img, self.feature_mask(disp), self.nfeatures)
        nxt = dict(img=img, disp=disp, d3=d3, kps=k, desc=k["desc"])
        if len(k["xy"]) < self.min_matches:
            self.skipped_frames += 1
            self.skip_cause = "keypoints"
            return False
        if self.cur is None:
            self.cur = nxt
            return True
        T = None
        a, b = self.point_clouds(self.cur, nxt)
        if a is None:
            self.skip_cause = "matches"
        else:
            T = self.point_cloud_transform(a, b)
            if T is not None:
                self.c_T_w_prev = self.c_T_w
                self.c_T_w = T @ self.c_T_w
        if T is None and self.prev is not None:
            a, b = self.point_clouds(self.prev, nxt)
            if a is None:
                self.skip_cause = "matches"
            else:
                T = self.point_cloud_transform(a, b)
                if T is not None:
                    T_prev = self.c_T_w_prev
                    self.c_T_w_prev = self.c_T_w
                    self.c_T_w = T @ T_prev
                    self.skipped_frames = 0
        if T is None:
            self.skipped_frames += 1
            return False
        self.skipped_frames = 0
        self.prev, self.cur = self.cur, nxt
        return True

    def point_clouds(self, f1, f2):                              # :162-175
        idx, dist = O.bf_knn2_hamming(f1["desc"], f2["desc"])
        q, t = O.ratio_filter(idx, dist, self.match_threshold)
        if len(q) < self.min_matches:
            return None, None
        p1, s1 = f1["d3"].sample(f1["kps"]["xy"][q])
        p2, s2 = f2["d3"].sample(f2["kps"]["xy"][t])
        if (s1 == 2).any() or (s2 == 2).any():
            raise ZeroDivisionError("division by zero")
        self.last_matches = (q, t)
        return p1, p2

    def _estimate(self, src, dst):
        T, _ = O.umeyama(src, dst, True)
        return np.vstack([T, [0, 0, 0, 1]])

    def point_cloud_transform(self, current_pts, next_pts):      # :177-223
        if self.rigidity_threshold > 0:
            m = O.rigid_clique(current_pts, next_pts, self.rigidity_threshold)
            current_pts, next_pts = current_pts[m > 0], next_pts[m > 0]
        rigidity_cause = False
        if len(current_pts) < 10:
            rigidity_cause = True
            self.skip_cause = "rigidity"
        if self.outlier_threshold > 0 and len(current_pts) >= 10:
            T = self._estimate(current_pts, next_pts)
            h_pts = np.hstack([next_pts, np.ones((len(next_pts), 1))])
            h_prev = np.hstack([current_pts, np.ones((len(current_pts), 1))])
            errors = np.array([np.linalg.norm(h_pts[i] - T @ h_prev[i]) / np.linalg.norm(h_pts[i])
                               for i in range(len(h_pts))])
            threshold = self.outlier_threshold + np.median(errors)
            current_pts, next_pts = current_pts[errors < threshold], next_pts[errors < threshold]
        if len(current_pts) < self.min_matches:
            if not rigidity_cause:
                self.skip_cause = "outlier"
            return None
        T = self._estimate(current_pts, next_pts)
        if np.isnan(T).any():
            self.skip_cause = "nan"
            return None
        disp = T[0:3, 3]
        rot = O.rodrigues(T[0:3, 0:3])
        lim = self.skipped_frames + 1
        big_d = np.linalg.norm(disp) > self.MAX_DISTANCE_CHANGE * lim
        big_r = np.linalg.norm(rot) > self.MAX_ROTATION_CHANGE * lim
        if big_d or big_r:
            if big_d:
                self.skip_cause = "bigdist"
            if big_r:
                self.skip_cause = "bigrot"
            return None
        return T

    def current_pose(self):                                      # :225-226
        return np.linalg.inv(self.c_T_w)
