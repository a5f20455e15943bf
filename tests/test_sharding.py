"""Multi-GPU host logic on CPU (BASELINE config 3): the socket transport of openvo_amd.sharding
(no torch, no GPU) and what sharding can break -- the history the reference's update() carries from frame to
frame (/root/reference/src/openVO/stereo_odometer.py:137-160,215-220).  Two / three shard processes run a
scripted ("fake frame") StereoOdometer: the REAL update() state machine of openvo_amd with stand-ins for the
camera, ORB and the pair solver, so every decision (which frames pair up, fallback, gate widening) is the
product's own."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from openvo_amd import calib, sharding

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _spawn(script, world, tmp_path, args=()):
    path = tmp_path / "worker.py"
    path.write_text(script)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), WORLD_SIZE=str(world), PYTHONPATH=ROOT)
    procs = [subprocess.Popen([sys.executable, str(path)] + [str(a) for a in args], env=dict(env, RANK=str(r), LOCAL_RANK=str(r)),
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(world)]
    outs = [p.communicate(timeout=300)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), "\n".join(outs)
    return outs


def test_shard_ranges_cover_everything():
    for n, world in [(256, 8), (10, 4), (7, 2), (3, 8)]:
        spans = [sharding.shard_range(n, r, world) for r in range(world)]
        assert spans[0][0] == 0 and spans[-1][1] == n
        assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
        assert max(b - a for a, b in spans) - min(b - a for a, b in spans) <= 1


_PRIMS = r'''
import os, sys
import numpy as np
from openvo_amd import sharding
assert "torch" not in sys.modules
g, dev = sharding.init_from_env()
r, w = g.rank, g.world
assert g.transport == "socket"                       # no GPU here: the sockets carry everything
parts = g.all_gather_bytes(bytes([r]) * (r + 1))     # ragged payloads
assert parts == [bytes([i]) * (i + 1) for i in range(w)]
g.barrier()
assert g.broadcast_bytes(b"id-%d" % r, root=1) == b"id-1"
assert g.all_reduce_max(1.5 + r) == 1.5 + (w - 1)
a = g.all_gather_f64(np.arange(4) + 10 * r)
assert a.shape == (w, 4) and np.array_equal(a[:, 0], 10 * np.arange(w))
n = 5
T = np.tile(np.eye(4), (n, 1, 1)); T[:, 2, 3] = -0.25 * (np.arange(r * n, r * n + n) + 1); ok = np.ones(n); ok[1] = r != 0
allT, allok = g.gather_relative(T, ok)
assert allT.shape == (n * w, 4, 4) and allok.shape == (n * w,)
assert np.allclose(allT[:, 2, 3], -0.25 * (np.arange(n * w) + 1))
assert allok.tolist() == [True, False, True, True, True] + [True] * (n * (w - 1))
poses = sharding.compose(allT, allok)
assert abs(poses[-1][2, 3] - 0.25 * (sum(range(1, n * w + 1)) - 2)) < 1e-12
assert "torch" not in sys.modules
g.close()
print("PRIMS_OK")
'''


def test_socket_group_primitives_three_ranks(tmp_path):
    outs = _spawn(_PRIMS, 3, tmp_path)
    assert all("PRIMS_OK" in o for o in outs)


# ---- a scripted odometer: openvo_amd's own update() with stand-ins around it ----------------------------------
_FAKE = r'''
import numpy as np
from openvo_amd import StereoOdometer, calib


def motions(n, seed=7):
    """True frame-to-frame motion M_k (frame k-1 -> k), k = 1..n-1; M_0 unused."""
    rng = np.random.default_rng(seed)
    out = [np.eye(4)]
    for k in range(1, n):
        T = np.eye(4)
        T[:3, :3] = calib.rodrigues_vec_to_mat(rng.normal(scale=0.01, size=3))
        T[:3, 3] = [rng.normal(scale=0.02), rng.normal(scale=0.01), -0.25 + rng.normal(scale=0.01)]
        out.append(T)
    return out


class FakeStereo:
    def compute_3d(self, left, right, preprocessed=False):
        k = int(left)                                       # a "frame" is just its index
        return ("xyz", k), np.full((2, 2), 10.0, np.float32), ("img", k)


class FakeOrb:
    def __init__(self, nkp):
        self.nkp = nkp

    def detectAndCompute(self, img, mask):
        k = img[1]
        return [None] * self.nkp.get(k, 50), ("desc", k)


class ScriptedOdometer(StereoOdometer):
    """The pair solver is scripted: T(a -> b) is the product of the true motions, or None for the frame
    pairs listed in `fail` (as if too few matches survived)."""

    def __init__(self, M, fail=(), nkp=None):
        super().__init__(FakeStereo(), preprocessed_frames=True)
        self.orb, self.M, self.fail = FakeOrb(nkp or {}), M, set(fail)
        self.pairs_tried = []

    def _try_pair(self, kps_a, desc_a, im3d_a, kps_b, desc_b, im3d_b):
        a, b = desc_a[1], desc_b[1]
        self.pairs_tried.append((a, b))
        if (a, b) in self.fail:
            self.skip_cause = "matches"
            return None
        T = np.eye(4)
        for k in range(a + 1, b + 1):
            T = self.M[k] @ T
        return self._gate_like_reference(T)

    def _gate_like_reference(self, T):
        # the reference's motion gate on the translation, widened by skipped_frames [stereo_odometer.py:215-218]
        if np.linalg.norm(T[:3, 3]) > self.MAX_DISTANCE_CHANGE * (self.skipped_frames + 1):
            self.skip_cause = "bigdist"
            return None
        return T


def run_frames(M, frames, fail=(), nkp=None):
    """-> (relative transforms, accept flags, c_T_w after every frame) for the frame indices given."""
    from openvo_amd import sharding
    odo = ScriptedOdometer(M, fail, nkp)
    rel, ok, chain = [], [], []
    for k in frames:
        before = odo.c_T_w
        a = odo.update(k, k)
        ok.append(bool(a))
        rel.append(sharding.relative_from_chain(before, odo.c_T_w) if a else np.eye(4))
        chain.append(odo.c_T_w.copy())
    return np.array(rel), np.array(ok), chain, odo
'''

_SHARD = _FAKE + r'''
import json, os, sys
from openvo_amd import sharding
assert "torch" not in sys.modules
case = json.loads(sys.argv[1])
n_local, fail, nkp = case["n_local"], [tuple(p) for p in case["fail"]], {int(k): v for k, v in case["nkp"].items()}
g, _ = sharding.init_from_env()
N = n_local * g.world
M = motions(N)
lo, hi = g.rank * n_local, (g.rank + 1) * n_local
frames = list(range(lo, hi)) if g.rank == 0 else list(range(lo - 1, hi))     # one halo frame
rel, ok, chain, odo = run_frames(M, frames, fail, nkp)
if g.rank > 0:
    rel, ok = rel[1:], ok[1:]                                                # the halo frame belongs to the previous shard
allT, allok = g.gather_relative(rel, ok)
if g.rank == 0:
    poses = sharding.compose(allT, allok)
    _, sok, schain, _ = run_frames(M, range(N), fail, nkp)                   # the sequential reference chain
    seq = np.array([np.linalg.inv(c) for c in schain])
    report = sharding.boundary_report(allok, n_local, g.world)
    print("RESULT " + json.dumps({"max_err": float(np.abs(poses - seq).max()), "accepted": allok.tolist(), "seq_accepted": sok.tolist(),
                                  "report": report}))
g.close()
'''


def _run_case(tmp_path, world, case):
    import json
    outs = _spawn(_SHARD, world, tmp_path, args=[json.dumps(case)])
    line = [l for l in outs[0].splitlines() if l.startswith("RESULT ")][0]
    return json.loads(line[7:])


def test_two_shards_compose_to_the_sequential_trajectory(tmp_path):
    """No frame rejected at a chunk boundary: the gathered relative transforms compose to exactly the chain a
    single sequential odometer builds -- also with a frame rejected INSIDE a shard (the next frame then pairs
    with the older one, in both runs) and a frame with too few keypoints."""
    res = _run_case(tmp_path, 2, {"n_local": 8, "fail": [[2, 3], [1, 3]], "nkp": {"12": 3}})
    assert res["accepted"] == res["seq_accepted"]
    assert res["accepted"][3] is False and res["accepted"][12] is False and sum(res["accepted"]) == 14
    assert res["report"] == [] and res["max_err"] < 1e-12


def test_three_shards_all_accepted(tmp_path):
    res = _run_case(tmp_path, 3, {"n_local": 5, "fail": [], "nkp": {}})
    assert all(res["accepted"]) and res["report"] == [] and res["max_err"] < 1e-12


def test_rejected_frame_at_the_chunk_boundary_is_detected_and_reported(tmp_path):
    """The case sharding cannot make exact: the last frame of shard 0 (the halo of shard 1) is rejected.
    Sequentially frame 8 pairs with frame 6; shard 1 starts from frame 7 as if it had been accepted.  The
    composed trajectory then differs -- and boundary_report says so, naming the shard and the frame."""
    res = _run_case(tmp_path, 2, {"n_local": 8, "fail": [[6, 7], [5, 7]], "nkp": {}})
    assert res["accepted"][7] is False and res["seq_accepted"][7] is False
    assert len(res["report"]) == 1 and res["report"][0]["shard"] == 1 and res["report"][0]["frame"] == 8
    assert "halo frame 7 was rejected" in res["report"][0]["cause"]
    assert res["max_err"] > 1e-3                      # the inexactness is real, which is why it must be reported


def test_first_step_of_a_shard_rejected_is_reported(tmp_path):
    """Frame 8 cannot pair with frame 7 but could with frame 6 (the fallback): the sequential run recovers
    through `prev`, shard 1 has no `prev` yet."""
    res = _run_case(tmp_path, 2, {"n_local": 8, "fail": [[7, 8]], "nkp": {}})
    assert res["seq_accepted"][8] is True and res["accepted"][8] is False
    assert [r["frame"] for r in res["report"]] == [8] and "fallback" in res["report"][0]["cause"]


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


def test_single_process_group_is_the_identity():
    g = sharding.Group(0, 1)
    T = np.tile(np.eye(4), (3, 1, 1))
    a, ok = g.gather_relative(T, [1, 0, 1])
    assert a.shape == (3, 4, 4) and ok.tolist() == [True, False, True]
    assert g.all_reduce_max(2.5) == 2.5 and g.all_gather_bytes(b"x") == [b"x"]
    b, ok2 = sharding.gather_relative(T, [1, 1, 1])
    assert b.shape == (3, 4, 4) and ok2.all()


# ---- communicator creation: every rank takes the same decision ------------------------------------------------
_VOTE = r'''
import ctypes, os, sys
from openvo_amd import sharding


class StubRccl:
    """vo_mgpu_* of a library whose communicator creation fails on one rank (argv[1]) -- no GPU, no RCCL involved."""
    def __init__(self, bad_rank):
        self.bad, self.destroyed, self.created = bad_rank, 0, 0
    def vo_mgpu_unique_id(self, ident):
        ident[0] = 42
        return 0
    def vo_mgpu_create(self, device, rank, world, ident, out):
        assert ident[0] == 42 and world == int(os.environ["WORLD_SIZE"])
        if rank == self.bad:
            return -3
        self.created += 1
        return 0
    def vo_mgpu_destroy(self, h):
        self.destroyed += 1


bad = int(sys.argv[1])
g = sharding.Group(int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]))
stub = StubRccl(bad)
took = g.attach_rccl(0, lib=stub)
assert took == (bad < 0), took                       # one rank failing makes EVERY rank stay on the sockets
assert g.transport == ("rccl" if bad < 0 else "socket")
if bad >= 0 and g.rank != bad:
    assert stub.created == 1 and stub.destroyed == 1  # a communicator that was created is given back
assert g.describe()["transport"] == g.transport and g.describe()["world"] == g.world
if bad >= 0:
    assert g.all_reduce_max(1.0 + g.rank) == float(g.world)   # the socket transport carries on
g._mgpu = None                                        # (the stub has no collectives to close)
g.close()
print("VOTE_OK")
'''


@pytest.mark.parametrize("bad", [1, 0, -1])
def test_rccl_attach_vote_falls_back_together(tmp_path, bad):
    """attach_rccl with a stub library: communicator creation failing on one rank (a peer, or rank 0) leaves all three
    ranks on the socket transport, creation succeeding everywhere switches all three."""
    outs = _spawn(_VOTE, 3, tmp_path, args=(bad,))
    assert all("VOTE_OK" in o for o in outs)


def test_rendezvous_key_is_shared_by_a_job_and_differs_between_jobs(monkeypatch):
    """The rendezvous file is named after MASTER_ADDR:MASTER_PORT and the launcher's run id, not after a parent pid:
    ranks started by different parents find each other, two jobs on one node do not collide."""
    monkeypatch.setenv("MASTER_ADDR", "127.0.0.1")
    monkeypatch.setenv("MASTER_PORT", "29511")
    monkeypatch.delenv("TORCHELASTIC_RUN_ID", raising=False)
    monkeypatch.delenv("VO_RUN_ID", raising=False)
    k0 = sharding.rendezvous_key()
    monkeypatch.setenv("TORCHELASTIC_RUN_ID", "job/A")
    ka = sharding.rendezvous_key()
    monkeypatch.setenv("TORCHELASTIC_RUN_ID", "job/B")
    kb = sharding.rendezvous_key()
    monkeypatch.setenv("MASTER_PORT", "29512")
    kc = sharding.rendezvous_key()
    assert len({k0, ka, kb, kc}) == 4 and all("/" not in k and str(os.getppid()) not in k.split("_")[-1:] for k in (k0, ka, kb, kc))


def test_more_ranks_than_gpus_is_refused_without_a_mask(monkeypatch):
    """init_from_env: ranks outnumbering the visible GPUs without a per-rank device mask is an error (shared hardware
    queues collapse the rate), unless the rehearsal switch is set."""
    monkeypatch.setenv("WORLD_SIZE", "1")
    monkeypatch.setenv("RANK", "0")
    monkeypatch.setattr(sharding, "device_count", lambda: 1)
    g, dev = sharding.init_from_env()                   # one rank, one GPU: fine
    g.close()

    class FakeGroup:
        def __init__(self, rank, world):
            self.rank, self.world, self.closed = rank, world, False
        def close(self):
            self.closed = True
    monkeypatch.setenv("WORLD_SIZE", "2")
    monkeypatch.setattr(sharding, "Group", FakeGroup)
    for v in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES", "VO_SHARE_GPU"):
        monkeypatch.delenv(v, raising=False)
    with pytest.raises(RuntimeError, match="refusing to share a GPU"):
        sharding.init_from_env()
    monkeypatch.setenv("VO_SHARE_GPU", "1")
    g, dev = sharding.init_from_env(want_rccl=False)
    assert dev == 0 and g.world == 2


def test_rank_to_device_binding_under_device_masks(monkeypatch):
    """LOCAL_RANK is an index into the devices a rank SEES, not a physical GPU number, and under a one-device-per-rank mask it is
    no device index at all.  Every combination a launcher can produce on one node, with stand-ins for the device count, the
    group and the RCCL attach (which records the device it was asked for)."""
    attached = []

    class FakeGroup:
        def __init__(self, rank, world):
            self.rank, self.world = rank, world
        def close(self):
            pass
        def attach_rccl(self, device):
            attached.append(device)
    monkeypatch.setattr(sharding, "Group", FakeGroup)
    for v in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES", "VO_SHARE_GPU", "VO_NO_RCCL", "LOCAL_WORLD_SIZE"):
        monkeypatch.delenv(v, raising=False)

    def bind(world, local, ndev, mask=None, share=None, rank=None, local_world=None):
        monkeypatch.setenv("WORLD_SIZE", str(world))
        monkeypatch.setenv("RANK", str(local if rank is None else rank))
        monkeypatch.setenv("LOCAL_RANK", str(local))
        for name, val in (("HIP_VISIBLE_DEVICES", mask), ("VO_SHARE_GPU", share), ("LOCAL_WORLD_SIZE", local_world)):
            if val is None:
                monkeypatch.delenv(name, raising=False)
            else:
                monkeypatch.setenv(name, str(val))
        monkeypatch.setattr(sharding, "device_count", lambda: ndev)
        del attached[:]
        g, dev = sharding.init_from_env()
        return dev, list(attached)

    # the driver's launch: 8 ranks, 8 visible GPUs, no mask: device = LOCAL_RANK, RCCL on that device
    assert [bind(8, r, 8) for r in range(8)] == [(r, [r]) for r in range(8)]
    # a node-wide mask naming OTHER physical GPUs (4,5,6,7 for four ranks): still visible device LOCAL_RANK (0..3)
    assert [bind(4, r, 4, mask="4,5,6,7") for r in range(4)] == [(r, [r]) for r in range(4)]
    # more visible devices than ranks: LOCAL_RANK
    assert bind(2, 1, 8) == (1, [1])
    # a per-rank mask of ONE device (rank r sees only physical GPU r): LOCAL_RANK 5 is NOT a device index -- device 0, RCCL attached
    assert [bind(8, r, 1, mask=str(r)) for r in (0, 5, 7)] == [(0, [0])] * 3
    # a per-rank mask of two devices for eight ranks (two ranks per pair of GPUs would be LOCAL_RANK % 2)
    assert bind(8, 5, 2, mask="2,3") == (1, [1])
    # RANK != LOCAL_RANK (second node of a two-node job would say RANK 11, LOCAL_RANK 3, LOCAL_WORLD_SIZE 8): the device follows LOCAL_RANK
    assert bind(16, 3, 8, rank=11, local_world=8) == (3, [3])
    # a rehearsal on one GPU (VO_SHARE_GPU): everybody on device 0, no RCCL -- with or without a mask
    assert bind(8, 5, 1, share=8) == (0, [])
    assert bind(8, 5, 1, mask="0", share=8) == (0, [])
    # no mask, fewer GPUs than ranks, nobody said "share": refused
    with pytest.raises(RuntimeError, match="refusing to share a GPU"):
        bind(8, 5, 4)


def test_bench_orchestration_with_eight_socket_ranks(tmp_path):
    """bench.py --gpus 8 as the driver launches it (one process per rank, RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*), with the
    GPU work replaced by stand-ins (tests/bench_standin.py): eight ranks render their frames at the same time (the synthetic
    generator's library is built once behind a lock), meet on the socket transport, take the max of every window over the
    ranks, gather their poses, and rank 0 prints ONE JSON line with the contract's fields, the window samples and the host
    cores available per rank."""
    import json
    world = 8
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), WORLD_SIZE=str(world), LOCAL_WORLD_SIZE=str(world),
               PYTHONPATH=ROOT, VO_NO_RCCL="1", VO_RUN_ID="standin-%d" % os.getpid())
    cmd = [sys.executable, os.path.join(ROOT, "tests", "bench_standin.py"), "--gpus", str(world), "--steps", "4", "--warmup", "2",
           "--repeats", "2", "--cpu-pairs", "0", "--no-post", "--no-other"]
    procs = [subprocess.Popen(cmd, env=dict(env, RANK=str(r), LOCAL_RANK=str(r)), stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
             for r in range(world)]
    outs = [p.communicate(timeout=600) for p in procs]
    assert all(p.returncode == 0 for p in procs), "\n".join(o[1][-2000:] for o in outs)
    lines = [l for l in outs[0][0].splitlines() if l.startswith("{")]
    assert len(lines) == 1 and all(not o[0].strip() for o in outs[1:])          # rank 0 prints the one line
    d = json.loads(lines[0])
    assert d["n_gpus"] == world and d["steps"] == 4 and d["warmup"] == 2 and d["scaling"] == "weak" and d["unit"] == "frame-pairs/s"
    assert len(d["window_values"]) == 2 and d["ms_per_step"] == pytest.approx(float(np.median(d["window_ms"])) / 4, rel=1e-3)
    assert d["window_clock_mhz"] == [2400.0, 2400.0]                      # rank 0's clock reading after each window
    assert d["value"] == pytest.approx(4 * world / (d["ms_per_step"] * 4 / 1e3), rel=1e-3)      # whole-job pairs over the max-over-ranks time
    assert d["frames"] == world * 8 and d["accepted_frames"] == world * 8 and d["shard_boundaries_inexact"] == []
    assert d["cores_per_rank"] == pytest.approx(len(os.sched_getaffinity(0)) / world, abs=0.01)
    assert d["rccl"]["transport"] == "socket" and d["rccl"]["world"] == world
