"""MonoOdometer: the monocular front end of BASELINE config 5 on the GPU.

NOT part of the reference (openVO is stereo only and has no RANSAC); it exists because
BASELINE.json's north star names an "RANSAC essential-matrix ... hypothesis-scoring loop" and config 5 sizes it
(1920x1080, 8000 ORB keypoints, 5000 hypotheses).  Per frame: the image goes into a device slot, ORB keypoints
and descriptors stay there, and ONE native call (vo_mono_pair) runs brute-force Hamming kNN-2 against the
previous frame's descriptors, the ratio test, and the essential-matrix RANSAC on the surviving correspondences
-- the host synchronises once and receives E, the inlier count and (for the pose) the inlier correspondences.
The relative pose (R, unit-length t) comes from E by the usual four-fold decomposition + cheirality vote on the
host (a dozen flops per inlier); the translation scale is unobservable, so the chained trajectory uses
|t| = 1 per accepted pair unless `scale` is supplied.

Parity: every device stage equals the build's own CPU restatement bit for bit (tests/test_gpu_configs.py);
there is no openVO oracle for this class.
"""
import os

import numpy as np

from . import _native


def _det3(M):
    return (M[0, 0] * (M[1, 1] * M[2, 2] - M[1, 2] * M[2, 1]) - M[0, 1] * (M[1, 0] * M[2, 2] - M[1, 2] * M[2, 0])
            + M[0, 2] * (M[1, 0] * M[2, 1] - M[1, 1] * M[2, 0]))


_W = np.array([[0.0, -1.0, 0.0], [1.0, 0.0, 0.0], [0.0, 0.0, 1.0]])


def decompose_essential(E):
    """The two rotations and the translation direction of an essential matrix (Hartley & Zisserman 9.6.2)."""
    U, _, Vt = np.linalg.svd(E)
    if _det3(U) < 0:
        U = -U
    if _det3(Vt) < 0:
        Vt = -Vt
    UW = U @ _W
    return UW @ Vt, (U @ _W.T) @ Vt, U[:, 2]


def recover_pose(E, x1, x2):
    """(R, t) with x2 ~ R x1 + t for normalised image points x1, x2 (n x 2): the candidate that puts most points
    in front of both cameras; ties go to the first of (R1, t), (R1, -t), (R2, t), (R2, -t).  Both depths change sign
    exactly when t does, so each rotation is evaluated once (with +t) and serves two candidates."""
    R1, R2, t = decompose_essential(E)
    n = len(x1)
    h1 = np.empty((n, 3)); h1[:, :2] = x1; h1[:, 2] = 1.0
    Rs = np.stack([R1, R2])                                       # (2, 3, 3)
    # depth of each point in camera 1 from the two rays: z1 (R h1) + t = z2 h2   (h2 = (x2, 1))
    a = np.matmul(h1[None], Rs.transpose(0, 2, 1))               # (2, n, 3): R h1
    hx, hy = x2[None, :, 0], x2[None, :, 1]
    ax, ay, az = a[..., 0], a[..., 1], a[..., 2]
    nx, ny, nz = hy * t[2] - t[1], t[0] - hx * t[2], hx * t[1] - hy * t[0]         # h2 x t
    dx, dy, dz = ay - az * hy, az * hx - ax, ax * hy - ay * hx                     # (R h1) x h2
    z1 = (nx * dx + ny * dy + nz * dz) / np.maximum(dx * dx + dy * dy + dz * dz, 1e-300)
    z2 = (z1 * ax + t[0]) * hx + (z1 * ay + t[1]) * hy + (z1 * az + t[2])          # x |h2|^2 > 0: only its sign counts
    plus = np.count_nonzero((z1 > 0) & (z2 > 0), axis=1)
    minus = np.count_nonzero((z1 < 0) & (z2 < 0), axis=1)
    good = (int(plus[0]), int(minus[0]), int(plus[1]), int(minus[1]))
    k = good.index(max(good))
    return (R1, R2)[k >> 1], (t if (k & 1) == 0 else -t), good[k]


class MonoOdometer:
    def __init__(self, K, img_size, nfeatures=8000, match_threshold=0.8, ransac_iters=5000, ransac_threshold=1.0,
                 min_inliers=30, seed=4321, device=0, context=None, solver=5):
        """K: 3x3 intrinsics; img_size = (width, height).  ransac_threshold is the Sampson distance in pixels.
        solver: 5 = five-point minimal solver (what cv2.findEssentialMat runs), 8 = eight-point."""
        if solver not in (5, 8):
            raise ValueError("solver is 5 or 8")
        self.solver = int(solver)
        K = np.asarray(K, np.float64)
        self.K, self.K4 = K, [K[0, 0], K[1, 1], K[0, 2], K[1, 2]]
        w, h = int(img_size[0]), int(img_size[1])
        self._own_ctx = context is None
        self._ctx = context or _native.Context(device, max(w, 64), max(h, 64), 16, int(nfeatures))
        self.nfeatures, self.match_threshold = int(nfeatures), float(match_threshold)
        self.ransac_iters, self.ransac_threshold, self.min_inliers, self.seed = int(ransac_iters), float(ransac_threshold), int(min_inliers), int(seed)
        self._c_T_w = np.eye(4)          # world (= first frame) expressed in the current camera frame, like StereoOdometer
        from concurrent.futures import ThreadPoolExecutor
        self._pool, self._pending = ThreadPoolExecutor(1), []      # accepted pairs' pose recoveries run beside the next pairs' GPU work
        self.last = None                 # dict of the last pair step
        self.skip_cause = ""
        # staged streams run ahead: the ORB extraction of the next frames is enqueued on look-ahead engines
        # (vo_prefetch_staged_mono) while the main stream matches and scores the current pair
        self.lookahead = int(os.environ.get("VO_MONO_LOOKAHEAD", "5"))
        # ... and so do the pair steps: frame k + 1's step against frame k is begun (vo_mono_pair_begin, own stream) before
        # frame k's own step has been collected, on the assumption that k will be accepted as the next reference.  The
        # chains of up to `speculate` later pairs overlap the current one's; a rejected frame voids them (they are collected
        # and dropped, the pair is run again against the reference that stayed).  Results do not depend on any of this.
        self.speculate = max(0, min(int(os.environ.get("VO_MONO_SPECULATE", "3")), _native.VO_NUM_MONO_ASYNC - 1))
        self._free = list(range(10))     # frame slots this odometer uses
        self._ahead = {}                 # staged index -> slot with its extraction in flight
        self._count = {}                 # slot -> keypoint count, once the host has collected it
        self._open = {}                  # (slot_a, slot_b) -> ticket of a step begun and not yet collected
        self._ref = None                 # (slot, xy) of the reference frame
        self._n_staged = 0
        self.speculation = {"begun": 0, "used": 0, "void": 0}

    def _void_open(self):
        """Collect and drop every begun step (their premise -- which frame is the reference -- fell)."""
        for key in list(self._open):
            self._ctx.mono_pair_end(self._open.pop(key), want_matches=True)
            self.speculation["void"] += 1

    def reset_lookahead(self):
        """Forget everything begun ahead (extractions and pair steps of frames not asked for yet): the next update starts from
        the reference frame alone.  For measurements: nothing computed before the clock starts is used after it."""
        self._drop_ahead()

    def restart(self):
        """Forget the reference frame too: the next update() is a first frame again (the pose accumulated so far stays)."""
        self._drop_ahead()
        if self._ref is not None:
            self._free.append(self._ref[0])
            self._ref = None
        self.last = None

    def _drop_ahead(self):
        self._void_open()
        for s in self._ahead.values():                            # the predictions are void (their extractions may still run:
            self._ctx.lookahead_drop(s)                           # the native side orders the slot's next use behind them)
            self._count.pop(s, None)
            self._free.append(s)
        self._ahead = {}

    def stage_frames(self, frames):
        """Keep a list of images resident in HBM; update(k) with an int then reads frame k from there."""
        self._drop_ahead()
        self._ctx.stage_pairs([(f, f) for f in frames])
        self._n_staged = len(frames)

    def _begin(self, a, b):
        t = self._ctx.mono_pair_begin(a, b, self.match_threshold, self.K4, self.ransac_iters, self.ransac_threshold, self.seed,
                                      want_matches=True, solver=self.solver)
        self._open[(a, b)] = t
        return t

    def update(self, img, scale=1.0):
        """One frame (an image, or the index of a staged one); True when a relative pose was accepted (always True
        for the very first frame)."""
        ctx = self._ctx
        staged = isinstance(img, (int, np.integer))
        if staged and int(img) in self._ahead:
            cur = self._ahead.pop(int(img))                       # extraction already running on an engine
        else:
            self._drop_ahead()                                    # out-of-order request
            cur = self._free.pop()
            self._count.pop(cur, None)
            if staged:
                ctx.load_staged_pair(cur, int(img), True)
            else:
                ctx.upload_mono(cur, np.asarray(img))
        if staged:
            for j in range(int(img) + 1, min(int(img) + 1 + self.lookahead, self._n_staged)):
                if j not in self._ahead and len(self._free) > 1:
                    nxt = self._free.pop()
                    self._count.pop(nxt, None)
                    ctx.prefetch_staged_mono(nxt, j, self.nfeatures)
                    self._ahead[j] = nxt
        n = self._count.pop(cur, None)
        if n is None:
            n = ctx.orb_slot_count(cur, self.nfeatures, 0)
        if n < 8:
            self.skip_cause = "keypoints"
            self._void_open()
            self._free.append(cur)
            return False
        if self._ref is None:
            self._void_open()
            self._ref = (cur, ctx.download_keypoints_xy(cur).astype(np.float64))
            return True
        prev, xy_prev = self._ref
        key = (prev, cur)
        # steps begun ahead stand as long as they lie on the chain reference -> this frame -> the frames predicted next
        chain, a = {key}, cur
        if staged:
            for j in range(int(img) + 1, int(img) + 1 + self.speculate):
                if j not in self._ahead:
                    break
                chain.add((a, self._ahead[j]))
                a = self._ahead[j]
        for k2 in [k2 for k2 in self._open if k2 not in chain]:
            ctx.mono_pair_end(self._open.pop(k2), want_matches=True)
            self.speculation["void"] += 1
        if key in self._open:
            self.speculation["used"] += 1
        else:
            self._begin(prev, cur)
        # the later pairs, on the assumption that every frame up to them is accepted: only what is already extracted (no wait)
        if staged and self.speculate:
            a, na = cur, n
            for j in range(int(img) + 1, int(img) + 1 + self.speculate):
                b = self._ahead.get(j)
                if b is None or len(self._open) > self.speculate:
                    break
                if b not in self._count:
                    if not ctx.slot_ready(b):
                        break
                    self._count[b] = ctx.orb_slot_count(b, self.nfeatures, 0)
                if na < 8 or self._count[b] < 8:
                    break
                if (a, b) not in self._open:
                    self._begin(a, b)
                    self.speculation["begun"] += 1
                a, na = b, self._count[b]
        r = ctx.mono_pair_end(self._open.pop(key), want_matches=True)
        self.last = r
        need = 6 if self.solver == 5 else 8
        if r["matches"] < need or r["best_count"] < self.min_inliers:
            self.skip_cause = "matches" if r["matches"] < need else "inliers"
            self._void_open()            # the steps begun ahead took this frame for the next reference
            self._free.append(cur)
            return False                 # the previous frame stays the reference
        inl = np.nonzero(r["mask"])[0][:512]                      # a few hundred inliers decide the cheirality vote
        xy_cur = r["xy_b"][:n].astype(np.float64)
        xa, xb = xy_prev[r["q"][inl]], xy_cur[r["t"][inl]]
        fx, fy, cx, cy = self.K4
        # E -> (R, t) is host arithmetic on a few hundred inliers: it runs on a worker thread while this thread is inside
        # the next frame's native calls (which release the GIL); c_T_w / current_pose() collect it
        while len(self._pending) >= 8:   # (bounded: a reader of c_T_w never finds more than a few poses to wait for)
            self._collect_one()
        self._pending.append((self._pool.submit(recover_pose, r["E"].copy(), (xa - [cx, cy]) / [fx, fy], (xb - [cx, cy]) / [fx, fy]), float(scale)))
        self._free.append(prev)          # the new frame becomes the reference; the old slot is reused
        self._ref = (cur, xy_cur)
        return True

    def _collect_one(self):
        fut, scale = self._pending.pop(0)
        R, t, _ = fut.result()
        T = np.eye(4)
        T[:3, :3], T[:3, 3] = R, t / max(np.linalg.norm(t), 1e-300) * scale
        self._c_T_w = T @ self._c_T_w

    def _flush(self):
        while self._pending:
            self._collect_one()

    @property
    def c_T_w(self):
        self._flush()
        return self._c_T_w

    @c_T_w.setter
    def c_T_w(self, value):
        self._flush()
        self._c_T_w = np.asarray(value, np.float64)

    def close(self):
        """Collect the pending pose, stop the worker thread and release the native context (if this object created it)."""
        self._void_open()
        self._flush()
        self._pool.shutdown(wait=True)
        if self._own_ctx:
            self._ctx.close()

    def current_pose(self):
        return np.linalg.inv(self.c_T_w)
