"""Deterministic synthetic "corridor" stereo sequences.

Bench / test input generator: ideal rectified rig, analytic ray cast, hash-noise texture
(seed 1234), ground-truth trajectory z = 0.25 k, x = 0.3 sin(0.05 k), yaw = 0.01 sin(0.07 k).
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "libvo_synth.so")


class _Cfg(ctypes.Structure):
    _fields_ = [("w", ctypes.c_int), ("h", ctypes.c_int), ("f", ctypes.c_double),
                ("cx", ctypes.c_double), ("cy", ctypes.c_double), ("baseline", ctypes.c_double),
                ("half_width", ctypes.c_double), ("z_far", ctypes.c_double),
                ("ground_y", ctypes.c_double), ("seed", ctypes.c_uint32)]


def build_synth(force=False):
    src = os.path.join(_HERE, "corridor.c")

    def stale():
        return force or not os.path.exists(_LIB) or os.path.getmtime(_LIB) < os.path.getmtime(src)

    if stale():
        # several ranks of one job may get here together: one builds (into a temporary name, renamed when complete, so that
        # nobody ever loads a half-written library), the others wait on the lock and find the library up to date
        import fcntl
        with open(_LIB + ".lock", "w") as lk:
            fcntl.flock(lk, fcntl.LOCK_EX)
            try:
                if stale():
                    tmp = "%s.%d.tmp" % (_LIB, os.getpid())
                    subprocess.check_call(["gcc", "-O2", "-fPIC", "-shared", "-fopenmp", "-o", tmp, src, "-lm"])
                    os.replace(tmp, _LIB)
            finally:
                fcntl.flock(lk, fcntl.LOCK_UN)
    return _LIB


_lib = None


def _load():
    global _lib
    if _lib is None:
        build_synth()
        _lib = ctypes.CDLL(_LIB)
        _lib.vo_corridor_render_pair.argtypes = [ctypes.POINTER(_Cfg), ctypes.c_int,
                                                 ctypes.c_void_p, ctypes.c_void_p]
        _lib.vo_corridor_render_pair.restype = None
    return _lib


# name -> (w, h, f, cx, cy, B, half_width, z_far, numDisparities)
CONFIGS = {
    "C1": (640, 480, 400.0, 320.0, 240.0, 0.12, 3.0, 60.0, 64),
    "C2": (1280, 720, 718.856, 640.0, 360.0, 0.537, 6.0, 400.0, 128),
    "C5": (1920, 1080, 1050.0, 960.0, 540.0, 0.537, 6.0, 400.0, 16),   # monocular: left images only
    "C4": (2048, 1536, 1400.0, 1024.0, 768.0, 0.537, 6.0, 400.0, 256),
    # reduced shape used by fast CPU tests (not a BASELINE config)
    "T0": (320, 240, 200.0, 160.0, 120.0, 0.25, 3.0, 60.0, 32),
}


class Corridor:
    """Synthetic rig + scene for one BASELINE config."""

    def __init__(self, name="C2", seed=1234):
        w, h, f, cx, cy, B, hw, zf, D = CONFIGS[name]
        self.name, self.w, self.h, self.f, self.cx, self.cy, self.B, self.D = name, w, h, f, cx, cy, B, D
        self._cfg = _Cfg(w, h, f, cx, cy, B, hw, zf, 1.65, seed)

    def pair(self, k):
        lib = _load()
        left = np.empty((self.h, self.w), np.uint8)
        right = np.empty((self.h, self.w), np.uint8)
        lib.vo_corridor_render_pair(ctypes.byref(self._cfg), int(k), left.ctypes.data, right.ctypes.data)
        return left, right

    def pairs(self, first, n, threads=None):
        """Frames first .. first+n-1, rendered on a few host threads (the renderer is pure C, re-entrant,
        and ctypes drops the GIL around it)."""
        import os
        from concurrent.futures import ThreadPoolExecutor
        threads = threads or max(1, min(16, (os.cpu_count() or 2) // 2, n))
        if threads == 1:
            return [self.pair(first + i) for i in range(n)]
        with ThreadPoolExecutor(threads) as ex:
            return list(ex.map(self.pair, range(first, first + n)))

    # calibration of the ideal rig in the shapes StereoCamera takes
    def K(self):
        return np.array([[self.f, 0, self.cx], [0, self.f, self.cy], [0, 0, 1]], np.float64)

    def dist(self):
        return np.zeros(5, np.float64)

    def rect_params(self):
        return {"R": np.eye(3), "T": np.array([-self.B, 0.0, 0.0])}

    def sgbm_params(self, mode=None):
        p = dict(minDisparity=0, numDisparities=self.D, blockSize=5, P1=200, P2=800, disp12MaxDiff=1,
                 preFilterCap=63, uniquenessRatio=10, speckleWindowSize=100, speckleRange=2)
        if mode is not None:
            p["mode"] = mode
        return p

    def Q(self):
        """Q of the ideal rig as cv2.stereoRectify returns it (CALIB_ZERO_DISPARITY)."""
        return np.array([[1, 0, 0, -self.cx], [0, 1, 0, -self.cy], [0, 0, 0, self.f],
                         [0, 0, 1.0 / self.B, 0]], np.float64)

    @staticmethod
    def gt_pose(k):
        """4x4 camera-to-world pose of frame k relative to frame 0 (world = frame-0 camera)."""
        def abs_pose(j):
            z, x, yaw = 0.25 * j, 0.3 * np.sin(0.05 * j), 0.01 * np.sin(0.07 * j)
            c, s = np.cos(yaw), np.sin(yaw)
            T = np.eye(4)
            T[:3, :3] = [[c, 0, s], [0, 1, 0], [-s, 0, c]]
            T[:3, 3] = [x, 0, z]
            return T
        return np.linalg.inv(abs_pose(0)) @ abs_pose(k)
