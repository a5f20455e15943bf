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
import numpy as np

from . import _native


def decompose_essential(E):
    """The two rotations and the translation direction of an essential matrix (Hartley & Zisserman 9.6.2)."""
    U, _, Vt = np.linalg.svd(E)
    if np.linalg.det(U) < 0:
        U = -U
    if np.linalg.det(Vt) < 0:
        Vt = -Vt
    W = np.array([[0.0, -1.0, 0.0], [1.0, 0.0, 0.0], [0.0, 0.0, 1.0]])
    return U @ W @ Vt, U @ W.T @ Vt, U[:, 2]


def recover_pose(E, x1, x2):
    """(R, t) with x2 ~ R x1 + t for normalised image points x1, x2 (n x 2): the candidate that puts most points
    in front of both cameras."""
    R1, R2, t = decompose_essential(E)
    best, arg = -1, None
    h1 = np.c_[x1, np.ones(len(x1))]
    h2 = np.c_[x2, np.ones(len(x2))]
    for R in (R1, R2):
        for tt in (t, -t):
            # depth of each point in camera 1 from the two rays: z1 (R h1) + t = z2 h2
            a = h1 @ R.T
            num = np.cross(h2, np.broadcast_to(tt, h2.shape))
            den = np.cross(a, h2)
            z1 = np.sum(num * den, 1) / np.maximum(np.sum(den * den, 1), 1e-300)
            z2 = np.sum((z1[:, None] * a + tt) * h2, 1) / np.sum(h2 * h2, 1)
            good = int(np.sum((z1 > 0) & (z2 > 0)))
            if good > best:
                best, arg = good, (R, tt)
    return arg[0], arg[1], best


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
        self._ctx = context or _native.Context(device, max(w, 64), max(h, 64), 16, int(nfeatures))
        self.nfeatures, self.match_threshold = int(nfeatures), float(match_threshold)
        self.ransac_iters, self.ransac_threshold, self.min_inliers, self.seed = int(ransac_iters), float(ransac_threshold), int(min_inliers), int(seed)
        self.c_T_w = np.eye(4)           # world (= first frame) expressed in the current camera frame, like StereoOdometer
        self._slot, self._have_prev = 0, False
        self.last = None                 # dict of the last pair step
        self.skip_cause = ""

    def stage_frames(self, frames):
        """Keep a list of images resident in HBM; update(k) with an int then reads frame k from there."""
        self._ctx.stage_pairs([(f, f) for f in frames])

    def update(self, img, scale=1.0):
        """One frame (an image, or the index of a staged one); True when a relative pose was accepted (always True
        for the very first frame)."""
        ctx, cur = self._ctx, self._slot
        if isinstance(img, (int, np.integer)):
            ctx.load_staged_pair(cur, int(img), True)
        else:
            ctx.upload_mono(cur, np.asarray(img))
        n = ctx.orb_slot_count(cur, self.nfeatures, 0)
        if n < 8:
            self.skip_cause = "keypoints"
            return False
        if not self._have_prev:
            self._have_prev, self._slot = True, 1 - cur
            return True
        prev = 1 - cur
        r = ctx.mono_pair(prev, cur, self.match_threshold, self.K4, self.ransac_iters, self.ransac_threshold, self.seed, want_matches=True, solver=self.solver)
        self.last = r
        need = 6 if self.solver == 5 else 8
        if r["matches"] < need or r["best_count"] < self.min_inliers:
            self.skip_cause = "matches" if r["matches"] < need else "inliers"
            return False                 # the previous frame stays the reference
        inl = np.nonzero(r["mask"])[0][:512]                      # a few hundred inliers decide the cheirality vote
        xa = ctx.download_keypoints(prev)["xy"][r["q"][inl]].astype(np.float64)
        xb = ctx.download_keypoints(cur)["xy"][r["t"][inl]].astype(np.float64)
        fx, fy, cx, cy = self.K4
        R, t, _ = recover_pose(r["E"], (xa - [cx, cy]) / [fx, fy], (xb - [cx, cy]) / [fx, fy])
        T = np.eye(4)
        T[:3, :3], T[:3, 3] = R, t / max(np.linalg.norm(t), 1e-300) * float(scale)
        self.c_T_w = T @ self.c_T_w
        self._slot = prev                # the new frame becomes the reference; the old slot is reused
        return True

    def current_pose(self):
        return np.linalg.inv(self.c_T_w)
