"""The reference-side binding, runnable: a module that offers exactly the slice of `cv2` the reference touches,
backed by libvo355 (HIP kernels behind the C ABI) and by openvo_amd.calib for the one-off rectification setup.

    import sys, openvo_amd.cv2_compat as shim
    sys.modules["cv2"] = shim                  # before `import openVO`
    from openVO import StereoCamera, StereoOdometer          # the reference's own, unmodified files

Every name below is one call site of the reference (paths relative to /root/reference/src/openVO/):

  stereo_camera.py:17-18   stereoRectify            -> calib.stereo_rectify (host, numpy)
  stereo_camera.py:19-22   initUndistortRectifyMap  -> calib.init_undistort_rectify_map (CV_16SC2 maps)
  stereo_camera.py:23-27   StereoSGBM_create        -> vo_set_sgbm + vo_sgbm_compute_host
  stereo_camera.py:30,33   remap                    -> vo_set_rectify_maps + vo_remap
  stereo_camera.py:45,47   cvtColor(COLOR_BGR2GRAY) -> vo_cvt_bgr2gray
  stereo_camera.py:52      reprojectImageTo3D       -> vo_reproject_to_3d
  stereo_odometer.py:22    ORB_create, BFMatcher.create(NORM_HAMMING)
  stereo_odometer.py:117   orb.detectAndCompute     -> vo_orb_detect_and_compute_host
  stereo_odometer.py:163   matcher.knnMatch(k=2)    -> vo_bf_knn2_hamming
  stereo_odometer.py:190,204  estimateAffine3D(force_rotation=True) -> vo_umeyama
  stereo_odometer.py:212   Rodrigues                -> vo_rodrigues
  utils/drawPoseOnImage.py:29-36  putText           -> no-op (no font rasteriser; the overlay is out of scope)

This is the object-level seam: host arrays cross the boundary on every call, one frame at a time, which is what
the reference's structure allows.  The device-resident pipeline (frames kept in HBM, look-ahead engines, fused
pose step) needs openvo_amd.StereoCamera / StereoOdometer, which keep the same API.  One process-wide context is
created lazily on the first call that knows an image size.
"""
import numpy as np

from . import _native, calib
from .features import BFMatcher as _BFMatcher, ORB as _ORB

# constants the reference passes (values as in OpenCV 4.x)
CV_16SC2 = 11
INTER_LINEAR = 1
COLOR_BGR2GRAY = 6
NORM_HAMMING = 6
FONT_HERSHEY_SIMPLEX = 0
__version__ = "vo355-compat (OpenCV 4.x semantics)"

_state = {"ctx": None, "geom": (0, 0, 0, 0), "device": 0}


def set_device(device):
    _state["device"] = int(device)


def _context(w, h, ndisp=16, max_kp=2000):
    """The shared context, recreated (rarely) when a larger image / disparity range / keypoint budget shows up."""
    cw, ch, cd, ck = _state["geom"]
    if _state["ctx"] is None or w > cw or h > ch or ndisp > cd or max_kp > ck:
        if _state["ctx"] is not None:
            _state["ctx"].close()
        geom = (max(w, cw, 64), max(h, ch, 64), max(((ndisp + 15) // 16) * 16, cd, 16), max(max_kp, ck))
        _state["ctx"] = _native.Context(_state["device"], *geom)
        _state["geom"] = geom
    return _state["ctx"]


class error(Exception):
    """cv2.error stand-in (estimateAffine3D raises it for colinear / too few points)."""


# ---- stereo_camera.py -----------------------------------------------------------------------------------------
def stereoRectify(cameraMatrix1, distCoeffs1, cameraMatrix2, distCoeffs2, imageSize, R, T, *args, **kw):
    return calib.stereo_rectify(cameraMatrix1, distCoeffs1, cameraMatrix2, distCoeffs2, imageSize, R, T)


def initUndistortRectifyMap(cameraMatrix, distCoeffs, R, newCameraMatrix, size, m1type):
    if m1type != CV_16SC2:
        raise error("only CV_16SC2 maps are implemented (what the reference asks for)")
    return calib.init_undistort_rectify_map(cameraMatrix, distCoeffs, R, newCameraMatrix, size)


class _StereoSGBM:
    def __init__(self, params, mode):
        self.params, self.mode = params, int(mode)

    def compute(self, left, right):
        left, right = np.ascontiguousarray(left, np.uint8), np.ascontiguousarray(right, np.uint8)
        ctx = _context(left.shape[1], left.shape[0], self.params["numDisparities"])
        ctx.set_sgbm(self.params, self.mode)
        return ctx.sgbm_compute_host(left, right)


def StereoSGBM_create(minDisparity=0, numDisparities=16, blockSize=3, P1=0, P2=0, disp12MaxDiff=0, preFilterCap=0,
                      uniquenessRatio=0, speckleWindowSize=0, speckleRange=0, mode=0):
    return _StereoSGBM(dict(minDisparity=minDisparity, numDisparities=numDisparities, blockSize=blockSize, P1=P1, P2=P2,
                            disp12MaxDiff=disp12MaxDiff, preFilterCap=preFilterCap, uniquenessRatio=uniquenessRatio,
                            speckleWindowSize=speckleWindowSize, speckleRange=speckleRange), mode)


def remap(src, map1, map2, interpolation, *args, **kw):
    if interpolation != INTER_LINEAR:
        raise error("only INTER_LINEAR is implemented (what the reference asks for)")
    src = np.asarray(src)
    h, w = map1.shape[:2]
    ctx = _context(max(w, src.shape[1]), max(h, src.shape[0]))
    ctx.set_rectify_maps(0, np.ascontiguousarray(map1, np.int16), np.ascontiguousarray(map2, np.uint16))
    if src.ndim == 3:
        return np.stack([ctx.remap(0, np.ascontiguousarray(src[..., c]), (h, w)) for c in range(src.shape[2])], -1)
    return ctx.remap(0, np.ascontiguousarray(src), (h, w))


def cvtColor(src, code):
    if code != COLOR_BGR2GRAY:
        raise error("only COLOR_BGR2GRAY is implemented (what the reference asks for)")
    src = np.ascontiguousarray(src, np.uint8)
    return _context(src.shape[1], src.shape[0]).cvt_bgr2gray(src)


def reprojectImageTo3D(disparity, Q, *args, **kw):
    d = np.ascontiguousarray(disparity, np.float32)
    return _context(d.shape[1], d.shape[0]).reproject_to_3d(d, np.asarray(Q, np.float64))


# ---- stereo_odometer.py ---------------------------------------------------------------------------------------
class _LazyORB:
    def __init__(self, nfeatures):
        self.nfeatures = int(nfeatures)

    def detectAndCompute(self, image, mask=None):
        img = np.asarray(image)
        ctx = _context(img.shape[1], img.shape[0], max_kp=self.nfeatures)
        return _ORB(ctx, self.nfeatures).detectAndCompute(img, mask)


def ORB_create(nfeatures=500, *args, **kw):
    return _LazyORB(nfeatures)


class BFMatcher:
    def __init__(self, normType=NORM_HAMMING, crossCheck=False):
        if normType != NORM_HAMMING or crossCheck:
            raise error("only NORM_HAMMING without cross-check is implemented (what the reference asks for)")

    @classmethod
    def create(cls, normType=NORM_HAMMING, crossCheck=False):
        return cls(normType, crossCheck)

    def knnMatch(self, queryDescriptors, trainDescriptors, k=2):
        q = np.asarray(queryDescriptors)
        n = max(len(q), 0 if trainDescriptors is None else len(trainDescriptors))
        ctx = _context(*_state["geom"][:3], max_kp=max(n, 16))
        return _BFMatcher(ctx).knnMatch(queryDescriptors, trainDescriptors, k)


def estimateAffine3D(src, dst, force_rotation=True):
    ctx = _context(*_state["geom"][:3])
    try:
        T, scale = ctx.umeyama(np.asarray(src, np.float32), np.asarray(dst, np.float32), bool(force_rotation))
    except _native.VoError as e:
        raise error(str(e))
    return T, scale


def Rodrigues(src):
    return _native.Context.rodrigues(np.asarray(src, np.float64)), None


def putText(img, text, org, fontFace, fontScale, color, thickness=1, *args, **kw):
    return img
