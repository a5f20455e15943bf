"""StereoCamera: drop-in for openVO's class of the same name (reference stereo_camera.py:6-55).

Same constructor, classmethod, attributes and methods; the per-pair work of compute_3d
(cvtColor -> remap x2 -> StereoSGBM -> reprojectImageTo3D -> crop) runs as HIP kernels on one
MI355X and its three return values are lazy device-backed array-likes (DeviceImage) that
download on first numpy use, so StereoOdometer can keep the whole frame on the GPU.
"""
import os
import pickle

import numpy as np

from . import _native, calib
from .features import DeviceImage, FrameHandle, StereoSGBM


class StagedPair:
    """Handle of an input pair already resident in HBM (see StereoCamera.stage_pairs); pass it as
    `img_left` (with img_right=None) to compute_3d / StereoOdometer.update."""

    def __init__(self, index):
        self.index = int(index)


class SubmittedPair:
    """Handle of a host pair handed to StereoCamera.submit(): its upload, disparity (and, once the
    odometer has announced its ORB settings, keypoints) already run on a look-ahead engine.  Pass it
    as `img_left` (with img_right=None) to compute_3d / StereoOdometer.update."""

    def __init__(self, slot, shape, preprocessed, images=None):
        self.slot, self.shape, self.preprocessed, self.images = slot, shape, bool(preprocessed), images


def _RESERVED():
    """Stand-in "owner" of a slot held by look-ahead work (callable like a weakref, never dead)."""
    return _RESERVED


def _load_plain(path):
    path = os.fspath(path)
    if path.lower().endswith(".json"):
        import json
        with open(path, "r") as fh:
            d = json.load(fh)
        if not isinstance(d, dict):
            raise ValueError("%s: expected a JSON object" % path)
        return d
    if path.lower().endswith(".npz"):
        with np.load(path, allow_pickle=False) as z:
            return {k: z[k] for k in z.files}
    raise ValueError("%s: expected a .json or .npz file (use from_pfiles for the reference's pickles)" % path)


def _save_plain(path, d):
    path = os.fspath(path)
    if path.lower().endswith(".json"):
        import json
        with open(path, "w") as fh:
            json.dump({k: (np.asarray(v).tolist() if not isinstance(v, (int, float)) else v) for k, v in d.items()}, fh)
    elif path.lower().endswith(".npz"):
        np.savez(path, **{k: np.asarray(v) for k, v in d.items()})
    else:
        raise ValueError("%s: expected a .json or .npz file" % path)


def _rotation_3x3(R):
    """'R' of a rectification file: a 3x3 matrix, or a rotation vector (3 numbers) converted like cv2.Rodrigues"""
    R = np.asarray(R, np.float64)
    if R.size == 9:
        return R.reshape(3, 3)
    if R.size != 3:
        raise ValueError("rectification 'R' must be a 3x3 matrix or a rotation vector of 3 numbers")
    return calib.rodrigues_vec_to_mat(R.reshape(3))


class StereoCamera:
    @classmethod
    def from_pfiles(cls, left_cam_file, right_cam_file, rect_file, sgbm_file, img_size, **kw):
        """Pickled dicts as the reference reads them (stereo_camera.py:8-14): cameras {'K','dist'},
        rectification {'R','T'}, and the ten StereoSGBM keys.  pickle.load executes arbitrary code:
        only use files you trust."""
        loaded = []
        for f in (left_cam_file, right_cam_file, rect_file, sgbm_file):
            with open(f, "rb") as fh:
                loaded.append(pickle.load(fh))
        cam_l, cam_r, rect, sgbm = loaded
        return cls(cam_l["K"], cam_l["dist"], cam_r["K"], cam_r["dist"], rect, sgbm, img_size, **kw)

    SGBM_KEYS = ("minDisparity", "numDisparities", "blockSize", "P1", "P2", "disp12MaxDiff", "preFilterCap",
                 "uniquenessRatio", "speckleWindowSize", "speckleRange")

    @classmethod
    def from_files(cls, left_cam_file, right_cam_file, rect_file, sgbm_file, img_size, **kw):
        """The same four dicts as from_pfiles from files that cannot execute code: each file is a .json
        object or an .npz archive (loaded with allow_pickle=False) holding the reference's keys -- cameras
        'K' (3x3) and 'dist', rectification 'R' (3x3 or rotation vector) and 'T', and the ten StereoSGBM
        integers.  Not in the reference (a safe alternative to pickle)."""
        cam_l, cam_r, rect, sgbm = (_load_plain(f) for f in (left_cam_file, right_cam_file, rect_file, sgbm_file))
        for name, d, keys in (("left camera", cam_l, ("K", "dist")), ("right camera", cam_r, ("K", "dist")),
                              ("rectification", rect, ("R", "T")), ("sgbm", sgbm, cls.SGBM_KEYS)):
            missing = [k for k in keys if k not in d]
            if missing:
                raise KeyError("%s file lacks %s" % (name, missing))
        sgbm = {k: int(np.asarray(sgbm[k]).reshape(-1)[0]) for k in cls.SGBM_KEYS}
        return cls(np.asarray(cam_l["K"], np.float64), np.asarray(cam_l["dist"], np.float64),
                   np.asarray(cam_r["K"], np.float64), np.asarray(cam_r["dist"], np.float64),
                   {"R": _rotation_3x3(rect["R"]), "T": np.asarray(rect["T"], np.float64)}, sgbm, img_size, **kw)

    @staticmethod
    def save_files(left_cam_file, right_cam_file, rect_file, sgbm_file, K_left, dist_left, K_right, dist_right,
                   rect_params, sgbm_params):
        """Write the four calibration files from_files reads (.json or .npz by extension)."""
        _save_plain(left_cam_file, {"K": K_left, "dist": dist_left})
        _save_plain(right_cam_file, {"K": K_right, "dist": dist_right})
        _save_plain(rect_file, {"R": rect_params["R"], "T": rect_params["T"]})
        _save_plain(sgbm_file, {k: int(sgbm_params[k]) for k in StereoCamera.SGBM_KEYS})

    def __init__(self, K_left, dist_left, K_right, dist_right, rect_params, sgbm_params, img_size,
                 device=0, max_keypoints=2000, context=None, engines=None, lookahead=None):
        """img_size = (width, height) as cv2 takes it.  Extra keyword arguments (not in the reference):
        device = HIP device index, max_keypoints = largest nfeatures an odometer may ask for, engines = how many look-ahead
        engines the context may create (default 16; each holds a full SGBM workspace from its first use on -- 0.95 GB at
        1280x720 / D = 128 -- so `engines=4` bounds the library at ~4 GB beside the frame slots), lookahead = how many staged
        pairs run ahead of update() (default 24, never more than engines + 8 when engines is given)."""
        w, h = int(img_size[0]), int(img_size[1])
        (R1, R2, P1, P2, self.Q, self.valid_region_left,
         self.valid_region_right) = calib.stereo_rectify(K_left, dist_left, K_right, dist_right, (w, h),
                                                         rect_params["R"], rect_params["T"])
        self.map_left_1, self.map_left_2 = calib.init_undistort_rectify_map(K_left, dist_left, R1, P1, (w, h))
        self.map_right_1, self.map_right_2 = calib.init_undistort_rectify_map(K_right, dist_right, R2, P2, (w, h))
        self.img_size = (w, h)
        D = int(sgbm_params["numDisparities"])
        self._ctx = context or _native.Context(device, max(w, 64), max(h, 64), max(16, ((D + 15) // 16) * 16),
                                               int(max_keypoints), engines=engines)
        if context is not None and engines is not None:
            context.set_engines(engines)
        self._ctx.set_rectify_maps(0, self.map_left_1, self.map_left_2)
        self._ctx.set_rectify_maps(1, self.map_right_1, self.map_right_2)
        self._ctx.set_Q(self.Q)
        # crop_to_valid_region_left slices rows vr[1]:vr[3], cols vr[0]:vr[2] -- it reads the
        # (x, y, w, h) ROI as (x0, y0, x1, y1); that behaviour is kept (stereo_camera.py:35-37)
        vr = self.valid_region_left
        self._ctx.set_roi(vr[0], vr[1], vr[2], vr[3])
        self.stereoSGBM = StereoSGBM(self._ctx, sgbm_params)
        self._slot_owner = [None] * _native.VO_NUM_SLOTS   # weak bookkeeping: FrameHandle per slot
        self._slot_gen = [0] * _native.VO_NUM_SLOTS        # bumped whenever a slot receives a new pair
        self._next_slot = 0
        # staged pairs: how many following pairs run their SGBM ahead (default 24, at most VO_NUM_SLOTS - 3 = 25, which leaves the
        # from-host path short of slots; env VO_LOOKAHEAD overrides).  24 against 18, five alternating runs each: steady rate
        # 2254 +- 2 against 2208 +- 3 pairs/s, 20-pair window 1767 +- 9 against 1750 +- 7
        self.lookahead = int(os.environ.get("VO_LOOKAHEAD", "24"))
        if lookahead is not None:
            self.lookahead = max(0, min(int(lookahead), _native.VO_NUM_SLOTS - 3))
        elif engines is not None and "VO_LOOKAHEAD" not in os.environ:
            self.lookahead = min(self.lookahead, self._ctx.set_engines(0) + 8)
        self._lookahead = []         # [((index, preprocessed), slot, (w, h))] of the pairs in flight
        self._n_staged = 0
        self.lookahead_stop = None   # staged pairs at or beyond this index are never started ahead (None: up to the last staged pair)

    # ---- slot bookkeeping -------------------------------------------------------------------
    def _release_slot(self, slot, frame):
        if self._slot_owner[slot] is not None and self._slot_owner[slot]() is frame:
            self._slot_owner[slot] = None

    def _free_slot(self):
        n = _native.VO_NUM_SLOTS
        for k in range(n):
            s = (self._next_slot + k) % n
            ref = self._slot_owner[s]
            if ref is None or ref() is None:
                self._next_slot = (s + 1) % n
                return s
        return None

    def _acquire_slot(self):
        import weakref
        n = _native.VO_NUM_SLOTS
        s = self._free_slot()
        if s is not None:
            return s, weakref
        if self._lookahead:
            # reclaim the slot held by an unconsumed look-ahead before evicting a live frame
            s = self._lookahead.pop()[1]
            self._drop(s)
            return s, weakref
        # every slot is still referenced by user code: move the oldest frame to host memory
        # (slots reserved by submit() hold work that cannot be redone and are never taken)
        s = self._next_slot
        for k in range(n):
            if self._slot_owner[(self._next_slot + k) % n] is not _RESERVED:
                s = (self._next_slot + k) % n
                break
        else:
            raise RuntimeError("every frame slot is held by a submitted pair")
        old = self._slot_owner[s]()
        if old is not None:
            old.evict()
        self._slot_owner[s] = None
        self._next_slot = (s + 1) % n
        return s, weakref

    def stage_pairs(self, pairs):
        """Upload (left, right) pairs once and keep them in HBM; returns StagedPair handles.
        Not in the reference: lets a caller overlap / amortise host-to-device ingest."""
        self._ctx.stage_pairs(pairs)
        for hit in self._lookahead:
            self._drop(hit[1])
        self._lookahead = []
        self._n_staged = len(pairs)
        return [StagedPair(i) for i in range(len(pairs))]

    def submit(self, img_left, img_right, preprocessed=False):
        """Start upload + disparity (+ keypoints) of a host pair on a look-ahead engine and return a
        SubmittedPair to hand to compute_3d / StereoOdometer.update later, in submission order.  The
        arrays may be reused as soon as this returns.  Not in the reference: it is the ingest step a
        caller puts in front of update() so that pair i+k uploads while pair i is being tracked.
        With no free slot (more than VO_NUM_SLOTS - 3 pairs outstanding) the pair is kept on the host
        and processed synchronously when consumed."""
        img_left, img_right = np.asarray(img_left), np.asarray(img_right)
        if img_left.ndim != img_right.ndim:
            if img_left.ndim == 3:
                img_left = self._ctx.cvt_bgr2gray(img_left)
            if img_right.ndim == 3:
                img_right = self._ctx.cvt_bgr2gray(img_right)
        held = sum(1 for o in self._slot_owner if o is _RESERVED)
        slot = self._free_slot() if held < _native.VO_NUM_SLOTS - 3 else None
        if slot is None:
            return SubmittedPair(None, None, preprocessed, (img_left.copy(), img_right.copy()))
        shape = self._ctx.prefetch_pair(slot, img_left, img_right, preprocessed)
        self._slot_gen[slot] += 1
        self._slot_owner[slot] = _RESERVED
        return SubmittedPair(slot, shape, preprocessed)

    def submit_staged(self, buf, w, h, ch, preprocessed):
        """Second half of submit() for a pair that a helper thread has already copied into pinned staging buffer `buf`
        (Context.host_stage_pair): upload + disparity (+ keypoints) start on a look-ahead engine, nothing is copied on
        this thread.  With no free slot the pair is taken back out of the staging buffer and processed synchronously when
        consumed, like submit() (the caller's own arrays may have been reused by then)."""
        held = sum(1 for o in self._slot_owner if o is _RESERVED)
        slot = self._free_slot() if held < _native.VO_NUM_SLOTS - 3 else None
        if slot is None:
            return SubmittedPair(None, None, preprocessed, self._ctx.host_stage_fetch(buf, w, h, ch))
        shape = self._ctx.prefetch_host_staged(slot, buf, w, h, ch, preprocessed)
        self._slot_gen[slot] += 1
        self._slot_owner[slot] = _RESERVED
        return SubmittedPair(slot, shape, preprocessed)

    def _drop(self, slot):
        """Give back the slot of a look-ahead pair nobody will consume (the native side stops counting it as in flight;
        whatever still runs on it is ordered before the slot's next use)."""
        self._slot_owner[slot] = None
        self._ctx.lookahead_drop(slot)

    def release_submitted(self, submitted):
        """Give back the slot of a SubmittedPair that will never be passed to compute_3d / update (idempotent)."""
        if isinstance(submitted, SubmittedPair):
            if submitted.slot is not None and self._slot_owner[submitted.slot] is _RESERVED:
                self._drop(submitted.slot)
            submitted.slot, submitted.images = None, None

    def slot_key(self, slot):
        """(slot, generation): identifies the pair a slot holds right now."""
        return (slot, self._slot_gen[slot])

    def next_lookahead_slots(self, k=2):
        """Device slots of the staged pairs expected next (oldest unconsumed look-ahead first)."""
        return [h[1] for h in self._lookahead[:k]]

    def reset_lookahead(self):
        """Drop look-ahead work that has been started but not consumed (its slots are released and
        the pairs will be recomputed when asked for)."""
        self._ctx.synchronize()
        for hit in self._lookahead:
            self._drop(hit[1])
        self._lookahead = []

    # ---- reference API ----------------------------------------------------------------------
    def undistort_rectify_left(self, img):
        return self._rectify(0, img)

    def undistort_rectify_right(self, img):
        return self._rectify(1, img)

    def _rectify(self, cam, img):
        img = np.asarray(img)
        if img.ndim == 3:   # cv2.remap handles each channel alike
            return np.stack([self._rectify(cam, img[..., c]) for c in range(img.shape[2])], -1)
        return self._ctx.remap(cam, img, (self.img_size[1], self.img_size[0]))

    def crop_to_valid_region_left(self, img):
        vr = self.valid_region_left
        return img[vr[1]: vr[3], vr[0]: vr[2]]

    def crop_to_valid_region_right(self, img):
        vr = self.valid_region_right
        return img[vr[1]: vr[3], vr[0]: vr[2]]

    def compute_3d(self, img_left, img_right, preprocessed=False):
        """-> (img_3d float32 HcxWcx3, disparity float32 HcxWc, img_left uint8 HcxWc), cropped;
        each is a DeviceImage (np.asarray(x) or x[...] materialises it)."""
        submitted = None
        if isinstance(img_left, SubmittedPair):
            submitted = img_left
            if submitted.slot is None:                       # no slot was free at submit(): host copy
                if submitted.images is None:
                    raise ValueError("this SubmittedPair has already been consumed")
                img_left, img_right = submitted.images
                preprocessed = submitted.preprocessed
                submitted.images = None
                submitted = None
        staged = isinstance(img_left, StagedPair)
        if not staged and submitted is None:
            img_left, img_right = np.asarray(img_left), np.asarray(img_right)
        if not staged and submitted is None and img_left.ndim != img_right.ndim:
            # the reference converts each image independently (stereo_camera.py:44-47)
            if img_left.ndim == 3:
                img_left = self._ctx.cvt_bgr2gray(img_left)
            if img_right.ndim == 3:
                img_right = self._ctx.cvt_bgr2gray(img_right)
        import weakref
        slot = None
        if submitted is not None:
            if self._slot_owner[submitted.slot] is not _RESERVED:
                raise ValueError("this SubmittedPair has already been consumed")
            slot, (w, h) = submitted.slot, submitted.shape
            submitted.slot = None
        if staged:
            # was this pair already ingested + SGBM'd on the look-ahead stream?
            key = (img_left.index, bool(preprocessed))
            while self._lookahead:
                hit = self._lookahead.pop(0)
                if hit[0] == key:
                    slot, (w, h) = hit[1], hit[2]
                    break
                self._drop(hit[1])                       # stale prediction: give the slot back
        if slot is None:
            slot, _ = self._acquire_slot()
            self._slot_gen[slot] += 1
            if staged:
                w, h = self._ctx.load_staged_pair(slot, img_left.index, preprocessed)
            else:
                w, h = self._ctx.upload_pair(slot, img_left, img_right, preprocessed)
            self._ctx.sgbm_compute(slot)
        vr = self.valid_region_left
        # numpy slice semantics (negative / oversized bounds clip)
        y0, y1, _ = slice(vr[1], vr[3]).indices(h)
        x0, x1, _ = slice(vr[0], vr[2]).indices(w)
        frame = FrameHandle(self, slot, w, h, (x0, y0, max(x1, x0), max(y1, y0)))
        self._slot_owner[slot] = weakref.ref(frame)
        if staged and self.lookahead:
            # start the NEXT staged pairs on the look-ahead streams: their disparity overlaps this
            # pair's ORB / matching / pose kernels (only into free slots -- never evict for a guess)
            have = {h[0][0] for h in self._lookahead}
            stop = self._n_staged if self.lookahead_stop is None else min(self._n_staged, int(self.lookahead_stop))
            for idx in range(img_left.index + 1, min(img_left.index + 1 + int(self.lookahead), stop)):
                if idx in have:
                    continue
                nxt = self._free_slot()
                if nxt is None:
                    break
                shape = self._ctx.prefetch_staged_pair(nxt, idx, preprocessed)
                self._slot_gen[nxt] += 1
                self._slot_owner[nxt] = _RESERVED
                self._lookahead.append(((idx, bool(preprocessed)), nxt, shape))
        return DeviceImage(frame, "xyz"), DeviceImage(frame, "disp"), DeviceImage(frame, "left")
