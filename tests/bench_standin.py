"""Test scaffolding (CPU only): run bench.py's rank orchestration -- rendezvous, per-rank frame rendering, windows, barriers,
max over ranks, pose gather, the JSON line -- with the GPU work replaced by stand-ins, so that an N-rank launch can be
rehearsed on a box without GPUs (tests/test_sharding.py::test_bench_orchestration_with_eight_socket_ranks).  The stand-ins
live HERE, not in bench.py or the package: the product has no CPU path.

    python tests/bench_standin.py --gpus 8 --steps 4 --warmup 2 --cpu-pairs 0 --no-post --no-other
"""
import os
import runpy
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import openvo_amd                                   # noqa: E402
from openvo_amd import sharding                     # noqa: E402
from openvo_amd.synth import Corridor               # noqa: E402


class _Ctx:
    def synchronize(self): pass
    def enable_timing(self, on=True, stages=None): pass
    def timings(self, reset=False): return {k: (0.0, 0) for k in ("upload", "sgbm_cost", "sgbm_agg", "sgbm_wta", "sgbm_post", "orb", "match", "pose", "knn")}
    def sgbm_sweep_status(self): return 0
    def sgbm_last_schedule(self): return 1
    def sgbm_last_geometry(self): return (1152 * 720 * 128, 3)
    def measure_copy(self, *a): return 1.0
    def shader_clock(self, *a): return 2400.0
    def close(self): pass


class _Staged:
    def __init__(self, i): self.index = i


class FakeCamera:
    def __init__(self, *a, **kw):
        self._ctx, self.lookahead, self.lookahead_stop, self.frames = _Ctx(), 18, None, []

    def stage_pairs(self, pairs):
        self.frames = [(np.asarray(l), np.asarray(r)) for l, r in pairs]
        assert all(l.shape == self.frames[0][0].shape and l.dtype == np.uint8 for l, _ in self.frames)
        return [_Staged(i) for i in range(len(pairs))]

    def reset_lookahead(self): pass


class FakeOdometer:
    """update() accepts every frame and chains a fixed forward motion: the composed trajectory is checkable."""
    def __init__(self, cam, **kw):
        self.cam, self.c_T_w, self.skipped_frames, self.first = cam, np.eye(4), 0, True

    def reset_lookahead(self): pass

    def update(self, staged, _):
        assert 0 <= staged.index < len(self.cam.frames)
        time.sleep(0.0005)
        if self.first:
            self.first = False
            return True
        T = np.eye(4)
        T[2, 3] = -0.25
        self.c_T_w = T @ self.c_T_w
        return True


openvo_amd.StereoCamera, openvo_amd.StereoOdometer = FakeCamera, FakeOdometer
sharding.device_count = lambda: int(os.environ.get("WORLD_SIZE", "1"))      # one (imaginary) GPU per rank
sys.argv = ["bench.py"] + sys.argv[1:]
runpy.run_path(os.path.join(ROOT, "bench.py"), run_name="__main__")
