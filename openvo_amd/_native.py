"""ctypes binding of libvo355.so (C ABI: include/vo355.h).

The library is built in-tree by `build_native()` (hipcc, gfx950).  There is no CPU fallback:
creating a `Context` without a usable HIP device raises `VoError`.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("VO355_LIB") or os.path.join(_HERE, "libvo355.so")   # VO355_LIB: A/B another build of the same ABI
_CSRC = os.path.join(_HERE, "csrc")

VO_NUM_SLOTS = 28
VO_NUM_HOST_STAGE = 20
VO_NUM_MONO_ASYNC = 5
VO_NUM_POSE_ASYNC = 8
SCHED_DIAG, SCHED_DIAG_RAGGED, SCHED_UNFUSED = 1, 2, 3
T_STAGES = ("upload", "sgbm_cost", "sgbm_agg", "sgbm_wta", "sgbm_post", "orb", "match", "pose", "knn")

# every symbol include/vo355.h declares (checked by tests/test_abi.py)
SYMBOLS = [
    "vo_create", "vo_destroy", "vo_last_error", "vo_device_name", "vo_synchronize", "vo_set_engines",
    "vo_set_rectify_maps", "vo_set_sgbm", "vo_set_Q", "vo_set_roi", "vo_upload_pair",
    "vo_stage_pairs_alloc", "vo_stage_pair", "vo_load_staged_pair", "vo_prefetch_staged_pair",
    "vo_set_lookahead_orb", "vo_prefetch_pair",
    "vo_sgbm_compute", "vo_sgbm_compute_host", "vo_download_disparity_f32", "vo_download_xyz",
    "vo_download_left", "vo_download_right", "vo_cvt_bgr2gray", "vo_remap", "vo_reproject_to_3d",
    "vo_orb_detect_and_compute", "vo_orb_detect_and_compute_host", "vo_slot_num_keypoints", "vo_download_keypoints",
    "vo_bf_knn2_hamming", "vo_ratio_filter", "vo_points3d_at", "vo_bilinear_at", "vo_point_clouds",
    "vo_pose_pair", "vo_pose_pair_begin", "vo_pose_pair_end", "vo_ransac_essential", "vo_ransac_essential5", "vo_ransac_pnp", "vo_umeyama", "vo_rigid_clique", "vo_rodrigues", "vo_enable_timing", "vo_get_timings",
    "vo_sgbm_last_geometry", "vo_host_stage_pair", "vo_host_stage_fetch", "vo_prefetch_host_staged", "vo_lookahead_depth", "vo_lookahead_drop", "vo_sgbm_last_schedule", "vo_measure_copy", "vo_measure_knn", "vo_shader_clock", "vo_sgbm_sweep_status", "vo_sgbm_sweep_stats",
    "vo_upload_mono", "vo_prefetch_staged_mono", "vo_mono_pair", "vo_mono_pair_begin", "vo_mono_pair_end", "vo_slot_ready", "vo_host_stage_begin", "vo_host_stage_wait",
    "vo_device_count", "vo_mgpu_unique_id", "vo_mgpu_create", "vo_mgpu_destroy", "vo_mgpu_info", "vo_mgpu_last_error",
    "vo_mgpu_gather_poses", "vo_mgpu_all_gather_f64", "vo_mgpu_all_reduce_max_f64",
]


class VoError(RuntimeError):
    """Raised for any non-zero status of the native library (code in .code)."""

    def __init__(self, code, msg):
        super().__init__("libvo355 error %d: %s" % (code, msg))
        self.code = code


class SweepTimeout(VoError):
    """VO_E_SWEEP: the disparity a result depends on is undefined (a strip hand-off of that pair's aggregation sweep gave up
    waiting, e.g. on an oversubscribed GPU).  Nothing computed from it is handed out; the pair can be submitted again."""


VO_E_SWEEP = -6


def _image_pair(left, right):
    """(left, right, channels) as C-contiguous uint8 arrays of one shape: HxW or HxWx3 (HxWx1 is squeezed); anything else
    -- e.g. HxWx4 -- is refused here, before a pointer reaches native code that would read w*h*channels bytes from it."""
    left, right = np.asarray(left), np.asarray(right)
    out = []
    for a in (left, right):
        if a.ndim == 3 and a.shape[2] == 1:
            a = a[:, :, 0]
        if a.ndim not in (2, 3) or (a.ndim == 3 and a.shape[2] != 3):
            raise ValueError("an image must be HxW or HxWx3 (got shape %s)" % (a.shape,))
        out.append(_c(a, np.uint8))
    if out[0].shape != out[1].shape:
        raise ValueError("left/right shapes differ")
    return out[0], out[1], (3 if out[0].ndim == 3 else 1)


def _image(img):
    """(img, channels) for ONE image under the same rule as _image_pair (monocular entry points)."""
    a = np.asarray(img)
    if a.ndim == 3 and a.shape[2] == 1:
        a = a[:, :, 0]
    if a.ndim not in (2, 3) or (a.ndim == 3 and a.shape[2] != 3):
        raise ValueError("an image must be HxW or HxWx3 (got shape %s)" % (a.shape,))
    return _c(a, np.uint8), (3 if a.ndim == 3 else 1)


def csrc_digest():
    """sha256 over the kernel sources (csrc/*.hip, *.inc, *.h + the public header), in name order: stamps a profile with the code it
    was taken from (tools/profile_round.sh) and lets bench.py say when the committed counters describe other kernels."""
    import hashlib
    h = hashlib.sha256()
    names = sorted(f for f in os.listdir(_CSRC) if f.endswith((".hip", ".inc", ".h")))
    for path in [os.path.join(_CSRC, f) for f in names] + [os.path.join(_HERE, "..", "include", "vo355.h")]:
        h.update(os.path.basename(path).encode() + b"\0")
        h.update(open(path, "rb").read())
    return h.hexdigest()[:16]


def build_native(force=False):
    """Compile openvo_amd/csrc/*.hip for gfx950 into openvo_amd/libvo355.so."""
    srcs = [os.path.join(_CSRC, f) for f in os.listdir(_CSRC) if f.endswith((".hip", ".h", ".inc"))]
    srcs.append(os.path.join(_HERE, "..", "include", "vo355.h"))
    srcs.append(os.path.join(_HERE, "..", "include", "vo_orb_pattern.inc"))
    newest = max(os.path.getmtime(s) for s in srcs)
    hooks = os.path.join(_HERE, "libvo355_hooks.so")       # test-only build with the failure-injection hook (tests load it by path)
    built = os.path.join(_HERE, "libvo355.so")
    if force or any(not os.path.exists(q) or os.path.getmtime(q) < newest for q in (built, hooks)):
        subprocess.check_call(["make", "-C", _CSRC, "-j4", "all"] + (["-B"] if force else []),
                              stdout=subprocess.DEVNULL)
    return LIB_PATH


_lib = None


def lib():
    """Load libvo355.so; raises if it has not been built (no silent fallback)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError("libvo355.so is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                              "(there is no CPU fallback)")
        L = ctypes.CDLL(LIB_PATH)
        L.vo_last_error.restype = ctypes.c_char_p
        L.vo_last_error.argtypes = [ctypes.c_void_p]
        L.vo_destroy.restype = None
        L.vo_destroy.argtypes = [ctypes.c_void_p]
        vp, ci, cd = ctypes.c_void_p, ctypes.c_int, ctypes.c_double
        L.vo_create.argtypes = [ci, ci, ci, ci, ci, ctypes.POINTER(vp)]
        L.vo_set_engines.argtypes = [vp, ci]
        L.vo_device_name.argtypes = [vp, ctypes.c_char_p, ci]
        L.vo_synchronize.argtypes = [vp]
        L.vo_set_rectify_maps.argtypes = [vp, ci, vp, vp, ci, ci]
        L.vo_set_sgbm.argtypes = [vp] + [ci] * 11
        L.vo_set_Q.argtypes = [vp, vp]
        L.vo_set_roi.argtypes = [vp, ci, ci, ci, ci]
        L.vo_upload_pair.argtypes = [vp, ci, vp, vp, ci, ci, ci, ci]
        L.vo_stage_pairs_alloc.argtypes = [vp, ci, ci, ci, ci]
        L.vo_stage_pair.argtypes = [vp, ci, vp, vp]
        L.vo_load_staged_pair.argtypes = [vp, ci, ci, ci]
        L.vo_prefetch_staged_pair.argtypes = [vp, ci, ci, ci]
        L.vo_set_lookahead_orb.argtypes = [vp, ci, ci, ci, ci, ci]
        L.vo_prefetch_pair.argtypes = [vp, ci, vp, vp, ci, ci, ci, ci]
        L.vo_sgbm_compute.argtypes = [vp, ci, vp]
        L.vo_sgbm_compute_host.argtypes = [vp, vp, vp, ci, ci, vp]
        for f in (L.vo_download_disparity_f32, L.vo_download_xyz, L.vo_download_left, L.vo_download_right):
            f.argtypes = [vp, ci, vp]
        L.vo_cvt_bgr2gray.argtypes = [vp, vp, ci, ci, vp]
        L.vo_remap.argtypes = [vp, ci, vp, ci, ci, vp]
        L.vo_reproject_to_3d.argtypes = [vp, vp, ci, ci, vp, vp]
        L.vo_orb_detect_and_compute.argtypes = [vp, ci, ci, ci, ci, ci, vp, vp, vp, vp, vp, vp, ci, vp]
        L.vo_orb_detect_and_compute_host.argtypes = [vp, vp, ci, ci, ci, vp, ci, ci, vp, vp, vp, vp, vp, vp, ci, vp]
        L.vo_slot_num_keypoints.argtypes = [vp, ci, vp]
        L.vo_download_keypoints.argtypes = [vp, ci, vp, vp, vp, vp, vp, vp, ci, vp]
        L.vo_bf_knn2_hamming.argtypes = [vp, vp, ci, vp, ci, vp, vp]
        L.vo_ratio_filter.argtypes = [vp, vp, ci, cd, vp, vp, vp]
        L.vo_points3d_at.argtypes = [vp, ci, vp, ci, vp, vp]
        L.vo_bilinear_at.argtypes = [vp, vp, ci, ci, vp, ci, vp, vp]
        L.vo_point_clouds.argtypes = [vp, ci, ci, cd, vp, vp, vp, vp, vp, vp, ci, vp]
        L.vo_pose_pair.argtypes = [vp, ci, ci, cd, ci, cd, cd, vp, vp, vp, vp]
        L.vo_pose_pair_begin.argtypes = [vp, ci, ci, cd, ci, cd, cd, vp]
        L.vo_pose_pair_end.argtypes = [vp, ci, vp, vp, vp, vp]
        L.vo_ransac_essential.argtypes = [vp, vp, vp, ci, vp, ci, ctypes.c_float, ctypes.c_uint32, vp, vp, vp, vp]
        L.vo_ransac_essential5.argtypes = L.vo_ransac_essential.argtypes
        L.vo_ransac_pnp.argtypes = [vp, vp, vp, ci, vp, ci, ctypes.c_float, ctypes.c_uint32, vp, vp, vp, vp]
        L.vo_umeyama.argtypes = [vp, vp, vp, ci, ci, vp, vp]
        L.vo_rigid_clique.argtypes = [vp, vp, vp, ci, cd, vp]
        L.vo_rodrigues.argtypes = [vp, vp]
        L.vo_enable_timing.argtypes = [vp, ci]
        L.vo_get_timings.argtypes = [vp, vp, vp, ci]
        L.vo_sgbm_last_geometry.argtypes = [vp, vp, vp]
        L.vo_sgbm_sweep_status.argtypes = [vp, vp]
        L.vo_lookahead_depth.argtypes = [vp, vp]
        L.vo_host_stage_pair.argtypes = [vp, ci, vp, vp, ci, ci, ci]
        L.vo_host_stage_fetch.argtypes = [vp, ci, vp, vp, ci, ci, ci]
        L.vo_prefetch_host_staged.argtypes = [vp, ci, ci, ci, ci, ci, ci]
        L.vo_lookahead_drop.argtypes = [vp, ci]
        L.vo_sgbm_last_schedule.argtypes = [vp, vp]
        L.vo_sgbm_sweep_stats.argtypes = [vp, ctypes.c_int, vp, ctypes.c_int]
        L.vo_measure_copy.argtypes = [vp, ctypes.c_int64, ci, ci, vp]
        L.vo_shader_clock.argtypes = [vp, ci, vp]
        L.vo_measure_knn.argtypes = [vp, ci, ci, ci, vp]
        L.vo_upload_mono.argtypes = [vp, ci, vp, ci, ci, ci]
        L.vo_prefetch_staged_mono.argtypes = [vp, ci, ci, ci]
        L.vo_mono_pair.argtypes = [vp, ci, ci, cd, vp, ci, ctypes.c_float, ctypes.c_uint32, ci, vp, vp, vp, vp, vp, ci]
        L.vo_slot_ready.argtypes = [vp, ci, vp]
        L.vo_host_stage_begin.argtypes = [vp, ci, vp, vp, ci, ci, ci]
        L.vo_host_stage_wait.argtypes = [vp, ci]
        L.vo_mono_pair_begin.argtypes = [vp, ci, ci, cd, vp, ci, ctypes.c_float, ctypes.c_uint32, ci, ci, vp]
        L.vo_mono_pair_end.argtypes = [vp, ci, vp, vp, vp, vp, vp, vp, ci]
        L.vo_device_count.argtypes = [vp]
        L.vo_mgpu_unique_id.argtypes = [vp]
        L.vo_mgpu_create.argtypes = [ci, ci, ci, vp, vp]
        L.vo_mgpu_destroy.restype = None
        L.vo_mgpu_destroy.argtypes = [vp]
        L.vo_mgpu_last_error.restype = ctypes.c_char_p
        L.vo_mgpu_last_error.argtypes = [vp]
        L.vo_mgpu_gather_poses.argtypes = [vp, vp, ci, vp]
        L.vo_mgpu_all_gather_f64.argtypes = [vp, vp, ci, vp]
        L.vo_mgpu_all_reduce_max_f64.argtypes = [vp, vp, ci]
        _lib = L
    return _lib


def _p(a):
    return None if a is None else a.ctypes.data_as(ctypes.c_void_p)


def _c(a, dt):
    return np.ascontiguousarray(a, dtype=dt)


class Context:
    """One device context (one HIP stream).  Not thread-safe: one per thread / per GPU."""

    def __init__(self, device=0, max_w=1280, max_h=720, max_disp=128, max_kp=2000, engines=None):
        """engines: how many look-ahead engines the context may use (None: the library's default of 16, or VO_ENGINES) -- each
        owns a stream and, from its first use on, a full SGBM + ORB workspace (0.95 GB at 1280x720 / D = 128)."""
        self._lib = lib()
        h = ctypes.c_void_p()
        rc = self._lib.vo_create(int(device), int(max_w), int(max_h), int(max_disp), int(max_kp), ctypes.byref(h))
        if rc != 0:
            raise VoError(rc, (self._lib.vo_last_error(None) or b"").decode())
        self._h = h
        if engines is not None:
            if int(engines) < 1:
                self.close()
                raise ValueError("engines must be >= 1")
            self.set_engines(engines)
        self._la_orb = None
        self.device, self.max_w, self.max_h, self.max_disp, self.max_kp = device, max_w, max_h, max_disp, max_kp
        self.kp_cap = max_kp * 2 + 1024

    def set_engines(self, n=0):
        """-> the number of look-ahead engines in effect (n <= 0 only asks)."""
        rc = self._lib.vo_set_engines(self._h, int(n))
        if rc < 0:
            raise VoError(rc, "vo_set_engines")
        return rc

    def close(self):
        if getattr(self, "_h", None):
            self._lib.vo_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _ck(self, rc):
        if rc != 0:
            raise (SweepTimeout if rc == VO_E_SWEEP else VoError)(rc, (self._lib.vo_last_error(self._h) or b"").decode())

    # ---- configuration
    def device_name(self):
        buf = ctypes.create_string_buffer(256)
        self._ck(self._lib.vo_device_name(self._h, buf, 256))
        return buf.value.decode()

    def synchronize(self):
        self._ck(self._lib.vo_synchronize(self._h))

    def set_rectify_maps(self, cam, map1, map2):
        map1, map2 = _c(map1, np.int16), _c(map2, np.uint16)
        h, w = map2.shape
        self._ck(self._lib.vo_set_rectify_maps(self._h, cam, _p(map1), _p(map2), w, h))

    def set_sgbm(self, p, mode=0):
        self._ck(self._lib.vo_set_sgbm(self._h, *[int(p[k]) for k in (
            "minDisparity", "numDisparities", "blockSize", "P1", "P2", "disp12MaxDiff", "preFilterCap",
            "uniquenessRatio", "speckleWindowSize", "speckleRange")], int(p.get("mode", mode))))

    def set_Q(self, Q):
        Q = _c(Q, np.float64)
        self._ck(self._lib.vo_set_Q(self._h, _p(Q)))

    def set_roi(self, x0, y0, x1, y1):
        self._ck(self._lib.vo_set_roi(self._h, int(x0), int(y0), int(x1), int(y1)))

    # ---- per pair
    def upload_pair(self, slot, left, right, preprocessed):
        left, right, ch = _image_pair(left, right)
        h, w = left.shape[:2]
        self._ck(self._lib.vo_upload_pair(self._h, slot, _p(left), _p(right), w, h, ch, int(bool(preprocessed))))
        return w, h

    def prefetch_pair(self, slot, left, right, preprocessed):
        """Look-ahead from host images: pinned staging + async upload + SGBM (+ ORB) on an engine."""
        left, right, ch = _image_pair(left, right)
        h, w = left.shape[:2]
        self._ck(self._lib.vo_prefetch_pair(self._h, slot, _p(left), _p(right), w, h, ch, int(bool(preprocessed))))
        return w, h

    def host_stage_pair(self, buf, left, right):
        """Copy a host pair into pinned staging buffer `buf` (waits for that buffer's previous upload).  The one call that
        may run on a helper thread while another thread drives this context."""
        left, right, ch = _image_pair(left, right)
        h, w = left.shape[:2]
        rc = self._lib.vo_host_stage_pair(self._h, int(buf), _p(left), _p(right), w, h, ch)
        if rc != 0:
            raise VoError(rc, "vo_host_stage_pair failed")      # (the context's error string belongs to the driving thread)
        return w, h, ch

    def host_stage_begin(self, buf, left, right):
        """The same copy on the library's own staging thread: returns at once with (w, h, ch, keep) -- `keep` holds the two
        arrays, which must stay alive and untouched until prefetch_host_staged / host_stage_fetch / host_stage_wait on `buf`."""
        left, right, ch = _image_pair(left, right)
        h, w = left.shape[:2]
        self._ck(self._lib.vo_host_stage_begin(self._h, int(buf), _p(left), _p(right), w, h, ch))
        return w, h, ch, (left, right)

    def host_stage_wait(self, buf):
        self._ck(self._lib.vo_host_stage_wait(self._h, int(buf)))

    def host_stage_fetch(self, buf, w, h, ch):
        shape = (h, w, 3) if ch == 3 else (h, w)
        left, right = np.empty(shape, np.uint8), np.empty(shape, np.uint8)
        self._ck(self._lib.vo_host_stage_fetch(self._h, int(buf), _p(left), _p(right), w, h, ch))
        return left, right

    def prefetch_host_staged(self, slot, buf, w, h, ch, preprocessed):
        self._ck(self._lib.vo_prefetch_host_staged(self._h, slot, int(buf), w, h, ch, int(bool(preprocessed))))
        return w, h

    def stage_pairs(self, pairs):
        """Upload a list of (left, right) host pairs once; they stay resident in HBM."""
        l0, _, ch = _image_pair(*pairs[0])
        h, w = l0.shape[:2]
        self._ck(self._lib.vo_stage_pairs_alloc(self._h, len(pairs), w, h, ch))
        for i, (l, r) in enumerate(pairs):
            l, r, _ = _image_pair(l, r)
            if l.shape != l0.shape:
                raise ValueError("all staged pairs must share one shape")
            self._ck(self._lib.vo_stage_pair(self._h, i, _p(l), _p(r)))
        self.staged_shape = (w, h)

    def load_staged_pair(self, slot, index, preprocessed):
        self._ck(self._lib.vo_load_staged_pair(self._h, slot, int(index), int(bool(preprocessed))))
        return self.staged_shape

    def lookahead_depth(self):
        """Look-ahead pairs submitted and not yet waited for or dropped."""
        d = ctypes.c_int(0)
        self._ck(self._lib.vo_lookahead_depth(self._h, ctypes.byref(d)))
        return d.value

    def lookahead_drop(self, slot):
        self._ck(self._lib.vo_lookahead_drop(self._h, int(slot)))

    def sgbm_last_schedule(self):
        """SCHED_* of the latest SGBM run of this context."""
        d = ctypes.c_int(0)
        self._ck(self._lib.vo_sgbm_last_schedule(self._h, ctypes.byref(d)))
        return d.value

    def prefetch_staged_pair(self, slot, index, preprocessed):
        self._ck(self._lib.vo_prefetch_staged_pair(self._h, slot, int(index), int(bool(preprocessed))))
        return self.staged_shape

    def sgbm_compute(self, slot, shape=None):
        out = np.empty(shape, np.int16) if shape is not None else None
        self._ck(self._lib.vo_sgbm_compute(self._h, slot, _p(out)))
        return out

    def sgbm_compute_host(self, left, right):
        left, right = _c(left, np.uint8), _c(right, np.uint8)
        h, w = left.shape
        out = np.empty((h, w), np.int16)
        self._ck(self._lib.vo_sgbm_compute_host(self._h, _p(left), _p(right), w, h, _p(out)))
        return out

    def download_disparity_f32(self, slot, shape):
        out = np.empty(shape, np.float32)
        self._ck(self._lib.vo_download_disparity_f32(self._h, slot, _p(out)))
        return out

    def download_xyz(self, slot, shape):
        out = np.empty(tuple(shape) + (3,), np.float32)
        self._ck(self._lib.vo_download_xyz(self._h, slot, _p(out)))
        return out

    def download_left(self, slot, shape, right=False):
        out = np.empty(shape, np.uint8)
        f = self._lib.vo_download_right if right else self._lib.vo_download_left
        self._ck(f(self._h, slot, _p(out)))
        return out

    def cvt_bgr2gray(self, bgr):
        bgr = _c(bgr, np.uint8)
        h, w = bgr.shape[:2]
        out = np.empty((h, w), np.uint8)
        self._ck(self._lib.vo_cvt_bgr2gray(self._h, _p(bgr), w, h, _p(out)))
        return out

    def remap(self, cam, src, out_shape):
        src = _c(src, np.uint8)
        out = np.empty(out_shape, np.uint8)
        self._ck(self._lib.vo_remap(self._h, cam, _p(src), src.shape[1], src.shape[0], _p(out)))
        return out

    def reproject_to_3d(self, disp, Q):
        disp, Q = _c(disp, np.float32), _c(Q, np.float64)
        out = np.empty(disp.shape + (3,), np.float32)
        self._ck(self._lib.vo_reproject_to_3d(self._h, _p(disp), disp.shape[1], disp.shape[0], _p(Q), _p(out)))
        return out

    # ---- features
    def _kp_buffers(self, cap):
        return dict(xy=np.empty((cap, 2), np.float32), size=np.empty(cap, np.float32),
                    angle=np.empty(cap, np.float32), response=np.empty(cap, np.float32),
                    octave=np.empty(cap, np.int32), desc=np.empty((cap, 32), np.uint8))

    @staticmethod
    def _trim(b, n):
        return {k: v[:n] for k, v in b.items()}

    def lookahead_orb(self, nfeatures, mask_mode, min_d16, max_d16):
        """Ask the look-ahead engines to extract keypoints with these parameters behind each prefetched SGBM."""
        key = (int(nfeatures), int(mask_mode), int(min_d16), int(max_d16))
        if key != self._la_orb:
            self._ck(self._lib.vo_set_lookahead_orb(self._h, 1, *key))
            self._la_orb = key

    def orb_slot(self, slot, nfeatures, mask_mode, min_d16=0, max_d16=0):
        cap = self.kp_cap
        b = self._kp_buffers(cap)
        n = ctypes.c_int(0)
        self._ck(self._lib.vo_orb_detect_and_compute(self._h, slot, int(nfeatures), int(mask_mode), int(min_d16),
                                                     int(max_d16), _p(b["xy"]), _p(b["size"]), _p(b["angle"]),
                                                     _p(b["response"]), _p(b["octave"]), _p(b["desc"]), cap,
                                                     ctypes.byref(n)))
        return self._trim(b, n.value)

    def orb_slot_count(self, slot, nfeatures, mask_mode, min_d16=0, max_d16=0):
        """Same extraction, everything stays on the device: returns only the keypoint count."""
        n = ctypes.c_int(0)
        self._ck(self._lib.vo_orb_detect_and_compute(self._h, slot, int(nfeatures), int(mask_mode), int(min_d16),
                                                     int(max_d16), None, None, None, None, None, None, 0, ctypes.byref(n)))
        return n.value

    def download_keypoints(self, slot):
        cap = self.kp_cap
        b = self._kp_buffers(cap)
        n = ctypes.c_int(0)
        self._ck(self._lib.vo_download_keypoints(self._h, slot, _p(b["xy"]), _p(b["size"]), _p(b["angle"]), _p(b["response"]),
                                                 _p(b["octave"]), _p(b["desc"]), cap, ctypes.byref(n)))
        return self._trim(b, n.value)

    def download_keypoints_xy(self, slot):
        """Only the (n, 2) float32 positions of a slot's keypoints (the other fields stay on the device)."""
        cap = self.kp_cap
        xy = np.empty((cap, 2), np.float32)
        n = ctypes.c_int(0)
        self._ck(self._lib.vo_download_keypoints(self._h, slot, _p(xy), None, None, None, None, None, cap, ctypes.byref(n)))
        return xy[:n.value]

    def orb_host(self, img, mask, nfeatures):
        img = np.asarray(img)
        if img.dtype != np.uint8 or img.ndim != 2:
            raise ValueError("ORB input must be a 2-D uint8 image")
        if img.strides[1] != 1:
            img = np.ascontiguousarray(img)
        h, w = img.shape
        mstride = 0
        if mask is not None:
            mask = np.asarray(mask, dtype=np.uint8)
            if mask.shape != img.shape:
                raise ValueError("mask shape differs from image shape")
            if mask.strides[1] != 1:
                mask = np.ascontiguousarray(mask)
            mstride = mask.strides[0]
        cap = self.kp_cap
        b = self._kp_buffers(cap)
        n = ctypes.c_int(0)
        self._ck(self._lib.vo_orb_detect_and_compute_host(self._h, _p(img), w, h, img.strides[0], _p(mask), mstride,
                                                          int(nfeatures), _p(b["xy"]), _p(b["size"]), _p(b["angle"]),
                                                          _p(b["response"]), _p(b["octave"]), _p(b["desc"]), cap,
                                                          ctypes.byref(n)))
        return self._trim(b, n.value)

    # ---- matching / 3-D / pose
    def bf_knn2(self, q, t):
        q, t = _c(q, np.uint8).reshape(-1, 32), _c(t, np.uint8).reshape(-1, 32)
        idx = np.empty((len(q), 2), np.int32)
        dist = np.empty((len(q), 2), np.int32)
        self._ck(self._lib.vo_bf_knn2_hamming(self._h, _p(q), len(q), _p(t), len(t), _p(idx), _p(dist)))
        return idx, dist

    def ratio_filter(self, idx, dist, ratio):
        idx, dist = _c(idx, np.int32), _c(dist, np.int32)
        qo, to = np.empty(len(idx), np.int32), np.empty(len(idx), np.int32)
        m = ctypes.c_int(0)
        rc = self._lib.vo_ratio_filter(_p(idx), _p(dist), len(idx), float(ratio), _p(qo), _p(to), ctypes.byref(m))
        if rc != 0:
            raise IndexError("list index out of range")  # what m[1] raises in the reference
        return qo[:m.value], to[:m.value]

    def points3d_at(self, slot, xy):
        xy = _c(xy, np.float32).reshape(-1, 2)
        out = np.empty((len(xy), 3), np.float32)
        st = np.empty(len(xy), np.uint8)
        self._ck(self._lib.vo_points3d_at(self._h, slot, _p(xy), len(xy), _p(out), _p(st)))
        return out, st

    def bilinear_at(self, img3d, xy):
        img3d, xy = _c(img3d, np.float32), _c(xy, np.float32).reshape(-1, 2)
        h, w = img3d.shape[:2]
        out = np.empty((len(xy), 3), np.float32)
        st = np.empty(len(xy), np.uint8)
        self._ck(self._lib.vo_bilinear_at(self._h, _p(img3d), w, h, _p(xy), len(xy), _p(out), _p(st)))
        return out, st

    def point_clouds(self, slot_a, slot_b, ratio):
        cap = self.kp_cap
        q, t = np.empty(cap, np.int32), np.empty(cap, np.int32)
        pa, pb = np.empty((cap, 3), np.float32), np.empty((cap, 3), np.float32)
        sa, sb = np.empty(cap, np.uint8), np.empty(cap, np.uint8)
        m = ctypes.c_int(0)
        self._ck(self._lib.vo_point_clouds(self._h, slot_a, slot_b, float(ratio), _p(q), _p(t), _p(pa), _p(pb),
                                           _p(sa), _p(sb), cap, ctypes.byref(m)))
        m = m.value
        return q[:m], t[:m], pa[:m], pb[:m], sa[:m], sb[:m]

    def pose_pair(self, slot_a, slot_b, ratio, min_matches, rigidity_thr, outlier_thr):
        """Fused match + ratio + 3-D lookup + clique filter + outlier pass + Umeyama for two slots.
        Returns (counts[M, n1, n2, flags], rc[first, final], T1 3x4, T2 3x4)."""
        counts = np.zeros(4, np.int32)
        rc = np.ones(2, np.int32)
        T1 = np.full((3, 4), np.nan)
        T2 = np.full((3, 4), np.nan)
        self._ck(self._lib.vo_pose_pair(self._h, int(slot_a), int(slot_b), float(ratio), int(min_matches),
                                        float(rigidity_thr), float(outlier_thr), _p(counts), _p(rc), _p(T1), _p(T2)))
        return counts, rc, T1, T2

    def pose_pair_begin(self, slot_a, slot_b, ratio, min_matches, rigidity_thr, outlier_thr):
        t = ctypes.c_int(-1)
        self._ck(self._lib.vo_pose_pair_begin(self._h, int(slot_a), int(slot_b), float(ratio), int(min_matches),
                                              float(rigidity_thr), float(outlier_thr), ctypes.byref(t)))
        return t.value

    def pose_pair_end(self, ticket):
        counts = np.zeros(4, np.int32)
        rc = np.ones(2, np.int32)
        T1 = np.full((3, 4), np.nan)
        T2 = np.full((3, 4), np.nan)
        self._ck(self._lib.vo_pose_pair_end(self._h, int(ticket), _p(counts), _p(rc), _p(T1), _p(T2)))
        return counts, rc, T1, T2

    def ransac_essential(self, pts1, pts2, K4, iters=5000, thr=1.0, seed=4321, want_counts=False, solver=8):
        if solver not in (5, 8):
            raise ValueError("solver is 5 (five-point) or 8 (eight-point)")
        pts1, pts2 = _c(pts1, np.float32).reshape(-1, 2), _c(pts2, np.float32).reshape(-1, 2)
        if len(pts1) != len(pts2):
            raise ValueError("point sets differ in length")
        K4 = _c(K4, np.float64)
        n = len(pts1)
        E = np.zeros(9, np.float64)
        mask = np.zeros(n, np.uint8)
        counts = np.zeros(iters, np.int32) if want_counts else None
        best = np.zeros(2, np.int32)
        fn = self._lib.vo_ransac_essential5 if solver == 5 else self._lib.vo_ransac_essential
        self._ck(fn(self._h, _p(pts1), _p(pts2), n, _p(K4), int(iters), float(thr), int(seed), _p(E), _p(mask), _p(counts), _p(best)))
        return dict(E=E.reshape(3, 3), mask=mask, counts=counts, best_iter=int(best[0]), best_count=int(best[1]))

    def upload_mono(self, slot, img):
        """One image into a slot (monocular front end)."""
        img, ch = _image(img)      # HxWx4 / HxWx2 are refused before native code would read w*h*3 bytes from them
        h, w = img.shape[:2]
        self._ck(self._lib.vo_upload_mono(self._h, slot, _p(img), w, h, ch))
        return w, h

    def prefetch_staged_mono(self, slot, index, nfeatures):
        self._ck(self._lib.vo_prefetch_staged_mono(self._h, int(slot), int(index), int(nfeatures)))

    def mono_pair(self, slot_a, slot_b, ratio, K4, iters=5000, thr=1.0, seed=4321, want_matches=False, solver=8):
        """kNN-2 + ratio + essential-matrix RANSAC between two slots' keypoints, all on the device, one sync.
        -> dict(E 3x3, matches M, best_iter, best_count[, mask, q, t of length M])."""
        K4 = _c(np.asarray(K4, np.float64).reshape(4), np.float64)
        E = np.zeros(9, np.float64)
        c3 = np.zeros(3, np.int32)
        cap = self.kp_cap
        mask = np.zeros(cap, np.uint8) if want_matches else None
        q = np.zeros(cap, np.int32) if want_matches else None
        t = np.zeros(cap, np.int32) if want_matches else None
        self._ck(self._lib.vo_mono_pair(self._h, int(slot_a), int(slot_b), float(ratio), _p(K4), int(iters), float(thr), int(seed) & 0xFFFFFFFF,
                                        int(solver), _p(E), _p(c3), _p(mask) if want_matches else None, _p(q) if want_matches else None,
                                        _p(t) if want_matches else None, cap))
        out = {"E": E.reshape(3, 3), "matches": int(c3[0]), "best_iter": int(c3[1]), "best_count": int(c3[2])}
        if want_matches:
            m = int(c3[0])
            out.update(mask=mask[:m].copy(), q=q[:m].copy(), t=t[:m].copy())
        return out

    def slot_ready(self, slot):
        r = ctypes.c_int(0)
        self._ck(self._lib.vo_slot_ready(self._h, int(slot), ctypes.byref(r)))
        return bool(r.value)

    def mono_pair_begin(self, slot_a, slot_b, ratio, K4, iters=5000, thr=1.0, seed=4321, want_matches=False, solver=8):
        """mono_pair in two halves (several pairs in flight): -> ticket for mono_pair_end."""
        K4 = _c(np.asarray(K4, np.float64).reshape(4), np.float64)
        t = ctypes.c_int(-1)
        self._ck(self._lib.vo_mono_pair_begin(self._h, int(slot_a), int(slot_b), float(ratio), _p(K4), int(iters), float(thr),
                                              int(seed) & 0xFFFFFFFF, int(solver), int(bool(want_matches)), ctypes.byref(t)))
        return t.value

    def mono_pair_end(self, ticket, want_matches=False):
        """-> the dict mono_pair returns (+ "xy_b": keypoint positions of the second slot, when want_matches)."""
        E = np.zeros(9, np.float64)
        c3 = np.zeros(3, np.int32)
        cap = self.kp_cap
        if want_matches:
            # one set of output arrays per context, reused call after call (the slices handed out are copies)
            buf = self.__dict__.get("_mono_out")
            if buf is None or len(buf[0]) != cap:
                buf = self._mono_out = (np.empty(cap, np.uint8), np.empty(cap, np.int32), np.empty(cap, np.int32), np.empty((cap, 2), np.float32))
            mask, q, t, xy = buf
            self._ck(self._lib.vo_mono_pair_end(self._h, int(ticket), _p(E), _p(c3), _p(mask), _p(q), _p(t), _p(xy), cap))
        else:
            self._ck(self._lib.vo_mono_pair_end(self._h, int(ticket), _p(E), _p(c3), None, None, None, None, cap))
        out = {"E": E.reshape(3, 3), "matches": int(c3[0]), "best_iter": int(c3[1]), "best_count": int(c3[2])}
        if want_matches:
            m = int(c3[0])
            out.update(mask=mask[:m].copy(), q=q[:m].copy(), t=t[:m].copy(), xy_b=xy.copy())
        return out

    def ransac_pnp(self, pts3d, pts2d, K4, iters=5000, thr=2.0, seed=4321, want_counts=False):
        pts3d, pts2d = _c(pts3d, np.float32).reshape(-1, 3), _c(pts2d, np.float32).reshape(-1, 2)
        if len(pts3d) != len(pts2d):
            raise ValueError("point sets differ in length")
        K4 = _c(K4, np.float64)
        n = len(pts3d)
        Rt = np.zeros(12, np.float64)
        mask = np.zeros(n, np.uint8)
        counts = np.zeros(iters, np.int32) if want_counts else None
        best = np.zeros(2, np.int32)
        self._ck(self._lib.vo_ransac_pnp(self._h, _p(pts3d), _p(pts2d), n, _p(K4), int(iters), float(thr), int(seed),
                                         _p(Rt), _p(mask), _p(counts), _p(best)))
        return dict(Rt=Rt.reshape(3, 4), mask=mask, counts=counts, best_iter=int(best[0]), best_count=int(best[1]))

    def umeyama(self, src, dst, force_rotation=True):
        src, dst = _c(src, np.float32).reshape(-1, 3), _c(dst, np.float32).reshape(-1, 3)
        if len(src) != len(dst):
            raise ValueError("Point sets need to have the same size")
        T = np.empty((3, 4), np.float64)
        s = ctypes.c_double(0)
        self._ck(self._lib.vo_umeyama(self._h, _p(src), _p(dst), len(src), int(bool(force_rotation)), _p(T),
                                      ctypes.byref(s)))
        return T, s.value

    def rigid_clique(self, prev, cur, thr):
        prev, cur = _c(prev, np.float32).reshape(-1, 3), _c(cur, np.float32).reshape(-1, 3)
        mask = np.zeros(len(cur), np.int64)
        self._ck(self._lib.vo_rigid_clique(self._h, _p(prev), _p(cur), len(cur), float(thr), _p(mask)))
        return mask

    @staticmethod
    def rodrigues(R):
        R = _c(R, np.float64)
        r = np.empty(3, np.float64)
        lib().vo_rodrigues(_p(R), _p(r))
        return r.reshape(3, 1)

    # ---- instrumentation
    def enable_timing(self, on=True, stages=None):
        """stages: iterable of stage names (T_STAGES) to restrict the event timing to."""
        flag = int(bool(on))
        if on and stages is not None:
            flag = sum(1 << T_STAGES.index(s) for s in stages) << 1
        self._ck(self._lib.vo_enable_timing(self._h, flag))

    def timings(self, reset=False):
        ms = np.zeros(len(T_STAGES), np.float64)
        n = np.zeros(len(T_STAGES), np.int64)
        self._ck(self._lib.vo_get_timings(self._h, _p(ms), _p(n), int(reset)))
        return {k: (float(ms[i]), int(n[i])) for i, k in enumerate(T_STAGES)}

    def sgbm_sweep_status(self):
        """Number of SGBM runs of this context whose diagonal sweep gave up a strip hand-off (0 = healthy; sticky).  Results that
        depend on such a run are refused with SweepTimeout where they are picked up."""
        e = ctypes.c_int(0)
        self._ck(self._lib.vo_sgbm_sweep_status(self._h, ctypes.byref(e)))
        return e.value

    def sgbm_sweep_stats(self, block=0, n_words=2048):
        """Control block of the latest aggregation sweep (development aid): see vo_sgbm_sweep_stats."""
        out = np.zeros(n_words, np.int32)
        self._ck(self._lib.vo_sgbm_sweep_stats(self._h, int(block), _p(out), n_words))
        return out

    def measure_copy(self, nbytes=0, reps=20, nontemporal=False):
        """GB/s (read + written) of a streaming device copy between two of the context's volumes."""
        g = ctypes.c_double(0.0)
        self._ck(self._lib.vo_measure_copy(self._h, int(nbytes), int(reps), 1 if nontemporal else 0, ctypes.byref(g)))
        return g.value

    def measure_knn(self, slot_a, slot_b, reps=20):
        """microseconds per launch of the Hamming kNN-2 kernel on two slots' descriptors (`reps` launches between two events)."""
        g = ctypes.c_double(0.0)
        self._ck(self._lib.vo_measure_knn(self._h, int(slot_a), int(slot_b), int(reps), ctypes.byref(g)))
        return g.value

    def shader_clock(self, micros=200):
        """MHz the shader clock holds right now (one wave counting cycles against the 100 MHz wall counter for `micros` us)."""
        g = ctypes.c_double(0.0)
        self._ck(self._lib.vo_shader_clock(self._h, int(micros), ctypes.byref(g)))
        return g.value

    def sgbm_last_geometry(self):
        cells, paths = ctypes.c_int64(0), ctypes.c_int(0)
        self._ck(self._lib.vo_sgbm_last_geometry(self._h, ctypes.byref(cells), ctypes.byref(paths)))
        return cells.value, paths.value
