"""Host-side logic that needs no GPU: rectification setup (sharding: tests/test_sharding.py)."""
import os

import numpy as np

from openvo_amd import calib
from openvo_amd.synth import Corridor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_stereo_rectify_ideal_rig():
    c = Corridor("C2")
    R1, R2, P1, P2, Q, roi1, roi2 = calib.stereo_rectify(c.K(), c.dist(), c.K(), c.dist(), (c.w, c.h),
                                                         c.rect_params()["R"], c.rect_params()["T"])
    assert np.allclose(R1, np.eye(3), atol=1e-12) and np.allclose(R2, np.eye(3), atol=1e-12)
    assert np.allclose(Q, c.Q(), atol=1e-9)
    assert abs(P2[0, 3] + c.f * c.B) < 1e-9
    assert roi1[0] == 0 and roi1[1] == 0 and roi1[2] >= c.w - 1 and roi1[3] >= c.h - 1
    m1, m2 = calib.init_undistort_rectify_map(c.K(), c.dist(), R1, P1, (c.w, c.h))
    xs, ys = np.meshgrid(np.arange(c.w), np.arange(c.h))
    assert np.array_equal(m1[..., 0], xs) and np.array_equal(m1[..., 1], ys) and (m2 == 0).all()


def test_rectify_rotated_rig_is_consistent():
    c = Corridor("C1")
    ang = 0.02
    R = calib.rodrigues_vec_to_mat([0.01, ang, -0.015])
    T = np.array([-c.B, 0.002, 0.001])
    dist = np.array([-0.1, 0.02, 0.001, -0.0005, 0.0])
    R1, R2, P1, P2, Q, roi1, roi2 = calib.stereo_rectify(c.K(), dist, c.K(), dist, (c.w, c.h), R, T)
    # after rectification the baseline lies along x: R2 * T is parallel to the x axis
    t = R2 @ T
    assert abs(t[1]) < 1e-9 and abs(t[2]) < 1e-9
    assert np.allclose(R2 @ R @ R1.T, np.eye(3), atol=1e-9)       # both cameras share one orientation
    assert 0 < roi1[2] <= c.w and 0 < roi1[3] <= c.h
    assert np.allclose(calib.rodrigues_mat_to_vec(calib.rodrigues_vec_to_mat([0.1, -0.2, 0.3])), [0.1, -0.2, 0.3])
    # undistort is the inverse of the distortion model
    pts = np.array([[100.0, 80.0], [500.0, 400.0]])
    n = calib.undistort_points(pts, c.K(), dist)
    k = calib._dist14(dist)
    x, y = n[:, 0], n[:, 1]
    r2 = x * x + y * y
    kr = 1 + ((k[4] * r2 + k[1]) * r2 + k[0]) * r2
    xd = x * kr + 2 * k[2] * x * y + k[3] * (r2 + 2 * x * x)
    yd = y * kr + k[2] * (r2 + 2 * y * y) + 2 * k[3] * x * y
    assert np.allclose(np.stack([xd * c.f + c.cx, yd * c.f + c.cy], 1), pts, atol=1e-6)


def _recover_pose_loop(E, x1, x2):
    """the four-candidate cheirality vote written out candidate by candidate (what mono.recover_pose computed before it
    evaluated each rotation once for both signs of t)"""
    from openvo_amd.mono import decompose_essential
    R1, R2, t = decompose_essential(E)
    best, arg = -1, None
    h1 = np.c_[x1, np.ones(len(x1))]
    h2 = np.c_[x2, np.ones(len(x2))]
    for R in (R1, R2):
        for tt in (t, -t):
            a = h1 @ R.T
            num = np.cross(h2, np.broadcast_to(tt, h2.shape))
            den = np.cross(a, h2)
            z1 = np.sum(num * den, 1) / np.maximum(np.sum(den * den, 1), 1e-300)
            z2 = np.sum((z1[:, None] * a + tt) * h2, 1) / np.sum(h2 * h2, 1)
            good = int(np.sum((z1 > 0) & (z2 > 0)))
            if good > best:
                best, arg = good, (R, tt)
    return arg[0], arg[1], best


def test_mono_recover_pose_known_motions_and_candidate_vote():
    """E -> (R, t) of the monocular odometer (host arithmetic): exact motions are recovered, and the shared evaluation of the
    (R, +t) / (R, -t) candidates picks the same candidate with the same vote as the candidate-by-candidate loop -- also with noisy
    points, points behind the camera and a pure rotation (t arbitrary: only the choice has to agree)."""
    from openvo_amd.mono import recover_pose
    rng = np.random.default_rng(11)

    def rot(v):
        return calib.rodrigues_vec_to_mat(np.asarray(v, np.float64))

    for trial in range(120):
        R = rot(rng.normal(size=3) * rng.uniform(0.01, 0.8))
        t = rng.normal(size=3)
        t /= np.linalg.norm(t)
        X = rng.uniform(-1, 1, (200, 3)) + [0, 0, rng.uniform(1.5, 6)]
        if trial % 5 == 4:
            X[:20, 2] *= -1                                            # a few points behind the first camera
        x1 = X[:, :2] / X[:, 2:]
        Y = X @ R.T + t
        x2 = Y[:, :2] / Y[:, 2:]
        if trial % 3 == 2:
            x2 = x2 + rng.normal(size=x2.shape) * 2e-3
        tx = np.array([[0, -t[2], t[1]], [t[2], 0, -t[0]], [-t[1], t[0], 0]])
        E = tx @ R
        Rg, tg, ng = recover_pose(E, x1, x2)
        Rl, tl, nl = _recover_pose_loop(E, x1, x2)
        assert ng == nl and np.array_equal(Rg, Rl) and np.array_equal(tg, tl), trial
        if trial % 3 != 2 and trial % 5 != 4:
            assert np.allclose(Rg, R, atol=1e-9) and np.allclose(tg, t, atol=1e-9) and ng > len(X) // 2, trial   # (a large rotation
            # leaves some points behind the second camera: they do not vote)
