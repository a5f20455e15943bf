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


class _FakeMonoContext:
    """Stands in for the native context under MonoOdometer: frames are their indices, a pair step's outcome is a function of the
    two frames it was begun on.  Records which steps were begun, collected and which slots were handed back."""
    kp_cap = 64

    def __init__(self, n_frames, few_keypoints=(), rejected=()):
        self.few, self.rejected = set(few_keypoints), set(rejected)
        self.frame_of, self.open, self.begun, self.ended = {}, {}, [], []
        self.next_ticket = 0

    def stage_pairs(self, pairs):
        self.n = len(pairs)

    def lookahead_drop(self, slot):
        pass

    def load_staged_pair(self, slot, idx, pre):
        self.frame_of[slot] = idx

    def prefetch_staged_mono(self, slot, idx, nf):
        self.frame_of[slot] = idx

    def slot_ready(self, slot):
        return True

    def orb_slot_count(self, slot, nf, mode):
        return 3 if self.frame_of[slot] in self.few else 40

    def download_keypoints_xy(self, slot):
        return np.zeros((40, 2), np.float32)

    def mono_pair_begin(self, a, b, *args, **kw):
        assert len(self.open) < 5, "more steps open than alternates"
        t = self.next_ticket
        self.next_ticket += 1
        self.open[t] = (self.frame_of[a], self.frame_of[b])
        self.begun.append(self.open[t])
        return t

    def mono_pair_end(self, ticket, want_matches=False):
        fa, fb = self.open.pop(ticket)
        self.ended.append((fa, fb))
        bad = fb in self.rejected
        m = 30
        return {"E": np.eye(3), "matches": m, "best_iter": fa * 100 + fb, "best_count": 2 if bad else 25,
                "mask": np.ones(m, np.uint8), "q": np.arange(m, dtype=np.int32), "t": np.arange(m, dtype=np.int32),
                "xy_b": np.zeros((self.kp_cap, 2), np.float32)}

    def close(self):
        pass


def test_mono_odometer_speculation_bookkeeping(monkeypatch):
    """MonoOdometer's host logic alone (a scripted context): with pair steps begun ahead, a frame with too few keypoints and
    a rejected frame void exactly the steps whose premise fell, every begun step is collected, and what update() reports --
    flags, causes, the pair each decision was taken on -- equals the run without speculation."""
    from openvo_amd import mono
    monkeypatch.setattr(mono, "recover_pose", lambda E, x1, x2: (np.eye(3), np.array([0.0, 0.0, 1.0]), len(x1)))
    out = {}
    for spec in (0, 3):
        ctx = _FakeMonoContext(12, few_keypoints={3}, rejected={6, 7})
        odo = mono.MonoOdometer(np.array([[100.0, 0, 32], [0, 100.0, 24], [0, 0, 1]]), (64, 48), nfeatures=40, min_inliers=10,
                                context=ctx)
        odo.speculate, odo.lookahead = spec, 5
        odo.stage_frames(list(range(12)))
        log = []
        for k in range(12):
            ok = odo.update(k)
            log.append((ok, "" if ok else odo.skip_cause, None if odo.last is None else odo.last["best_iter"]))
        odo.close()
        assert not ctx.open, "steps left open"
        assert sorted(ctx.begun) == sorted(ctx.ended)
        out[spec] = (log, dict(odo.speculation), list(ctx.begun), odo.c_T_w.copy())
    assert out[0][0] == out[3][0]
    assert np.array_equal(out[0][3], out[3][3])
    flags = [x[0] for x in out[3][0]]
    assert flags == [True, True, True, False, True, True, False, False, True, True, True, True]
    assert out[3][0][3][1] == "keypoints" and out[3][0][6][1] == "inliers" and out[3][0][7][1] == "inliers"
    # decisions were taken on (reference, frame): 2 stays the reference over the frame with few keypoints, 5 over the rejected ones
    assert [x[2] for x in out[3][0]][4] == 2 * 100 + 4 and [x[2] for x in out[3][0]][8] == 5 * 100 + 8
    assert out[0][1] == {"begun": 0, "used": 0, "void": 0}
    s = out[3][1]
    assert s["used"] >= 5 and s["void"] >= 2 and len(out[3][2]) > len(out[0][2])


def test_shared_gpu_queue_budget_never_exceeds_the_hardware_queues():
    """VO_SHARE_GPU=N: N processes' shares of the ~24 hardware queues must sum to <= 24 for every N up to 8 (they used to ask
    for 6 each from N = 5 on: 8 ranks rehearsed on one GPU requested 48), and each share still holds a main stream, a pose
    stream and an engine; N > 8 warns.  Runs `import openvo_amd` in a child process per N (the budget is read at import)."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import os, json, warnings\n"
            "with warnings.catch_warnings(record=True) as w:\n"
            "    warnings.simplefilter('always')\n"
            "    import openvo_amd\n"
            "print(json.dumps(dict(q=os.environ.get('GPU_MAX_HW_QUEUES'), e=os.environ.get('VO_ENGINES'), p=os.environ.get('VO_POSE_STREAMS'),"
            " la=os.environ.get('VO_LOOKAHEAD'), warned=len(w))))")
    for n in (0, 2, 3, 4, 5, 6, 8, 12):
        env = {k: v for k, v in os.environ.items() if k not in ("GPU_MAX_HW_QUEUES", "VO_ENGINES", "VO_POSE_STREAMS", "VO_LOOKAHEAD")}
        env["VO_SHARE_GPU"] = str(n)
        r = subprocess.run([sys.executable, "-c", code], cwd=root, env=env, capture_output=True, text=True, timeout=120)
        assert r.returncode == 0, r.stderr[-2000:]
        d = json.loads(r.stdout.strip().splitlines()[-1])
        q = int(d["q"])
        if n < 2:
            assert q == 24 and d["e"] is None
            continue
        e, p = int(d["e"]), int(d["p"])
        assert e >= 1 and p >= 1 and 1 + p + e <= q, d
        if n <= 8:
            assert n * q <= 24 and d["warned"] == 0, (n, d)
        else:
            assert d["warned"] == 1, (n, d)
