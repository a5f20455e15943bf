"""Host-side logic that needs no GPU: rectification setup, sharding + pose gather (gloo, 2 ranks)."""
import os
import subprocess
import sys

import numpy as np

from openvo_amd import calib, sharding
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


def test_shard_ranges_cover_everything():
    for n, world in [(256, 8), (10, 4), (7, 2), (3, 8)]:
        spans = [sharding.shard_range(n, r, world) for r in range(world)]
        assert spans[0][0] == 0 and spans[-1][1] == n
        assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
        assert max(b - a for a, b in spans) - min(b - a for a, b in spans) <= 1


def test_compose_matches_sequential_chain():
    rng = np.random.default_rng(0)
    c_T_w = np.eye(4)
    rel, ok, poses = [], [], []
    for k in range(12):
        T = np.eye(4)
        T[:3, :3] = calib.rodrigues_vec_to_mat(rng.normal(scale=0.01, size=3))
        T[:3, 3] = rng.normal(scale=0.1, size=3)
        acc = k % 5 != 3
        before = c_T_w.copy()
        if acc:
            c_T_w = T @ c_T_w
            assert np.allclose(sharding.relative_from_chain(before, c_T_w), T, atol=1e-12)
        rel.append(T); ok.append(acc); poses.append(np.linalg.inv(c_T_w))
    assert np.allclose(sharding.compose(rel, ok), poses, atol=1e-12)


_WORKER = r'''
import os, sys
sys.path.insert(0, %r)
import numpy as np, torch.distributed as dist
from openvo_amd import sharding
dist.init_process_group("gloo", rank=int(os.environ["RANK"]), world_size=int(os.environ["WORLD_SIZE"]))
r, w = dist.get_rank(), dist.get_world_size()
n = 5
lo, hi = sharding.shard_range(n * w, r, w)
T = np.tile(np.eye(4), (n, 1, 1)); T[:, 2, 3] = -0.25 * (np.arange(lo, hi) + 1); ok = np.ones(n); ok[1] = r
allT, allok = sharding.gather_relative(T, ok, dist)
if r == 0:
    assert allT.shape == (n * w, 4, 4) and allok.shape == (n * w,)
    assert np.allclose(allT[:, 2, 3], -0.25 * (np.arange(n * w) + 1))
    assert allok.tolist() == [True, False, True, True, True, True, True, True, True, True]
    poses = sharding.compose(allT, allok)
    assert abs(poses[-1][2, 3] - 0.25 * (sum(range(1, 11)) - 2)) < 1e-12
    print("GATHER_OK")
dist.barrier(); dist.destroy_process_group()
'''


def test_pose_gather_two_ranks_gloo(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(_WORKER % ROOT)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29577", WORLD_SIZE="2")
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r)), stdout=subprocess.PIPE,
                              stderr=subprocess.STDOUT, text=True) for r in range(2)]
    outs = [p.communicate(timeout=300)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    assert "GATHER_OK" in outs[0]
