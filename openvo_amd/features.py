"""Host-side stand-ins for the cv2 objects openVO holds as attributes.

The reference keeps `self.orb`, `self.matcher` (stereo_odometer.py:22) and `self.stereoSGBM`
(stereo_camera.py:23) and calls five methods on them; those call sites are the drop-in seam
(SURVEY.md section 8(b)).  The classes below expose the same methods and route them to the HIP
library.  Large arrays stay on the device behind lazy array-likes (`DeviceImage` & co.) and are
only downloaded when host code touches them.
"""
import numpy as np

MIN_KP_FIELDS = ("pt", "size", "angle", "response", "octave", "class_id")


class KeyPoint:
    """The fields of cv2.KeyPoint the reference reads (.pt at stereo_odometer.py:44-45,170-171)."""
    __slots__ = MIN_KP_FIELDS

    def __init__(self, x, y, size, angle=-1.0, response=0.0, octave=0, class_id=-1):
        self.pt = (float(x), float(y))
        self.size, self.angle, self.response = float(size), float(angle), float(response)
        self.octave, self.class_id = int(octave), int(class_id)

    def __repr__(self):
        return "KeyPoint(pt=%r, octave=%d, angle=%.3f, response=%g)" % (self.pt, self.octave, self.angle, self.response)


class DMatch:
    """cv2.DMatch fields used at stereo_odometer.py:164,166,170-171."""
    __slots__ = ("queryIdx", "trainIdx", "imgIdx", "distance")

    def __init__(self, queryIdx, trainIdx, distance, imgIdx=0):
        self.queryIdx, self.trainIdx, self.imgIdx, self.distance = int(queryIdx), int(trainIdx), int(imgIdx), float(distance)

    def __repr__(self):
        return "DMatch(q=%d, t=%d, d=%g)" % (self.queryIdx, self.trainIdx, self.distance)


class KeyPointList:
    """Sequence of KeyPoint backed by arrays (no per-keypoint Python objects until indexed).  A list that
    came out of the fused device path holds only its length until an array is asked for: the odometer's
    own matching / pose steps read keypoints and descriptors in the frame's device slot."""

    _FIELDS = ("xy", "size", "angle", "response", "octave")

    def __init__(self, arrays=None, frame=None, n=None):
        self._arr = arrays
        self._n = len(arrays["xy"]) if arrays is not None else int(n)
        self.frame = frame          # FrameHandle whose slot holds these keypoints on the device
        self.desc = None            # the descriptor array returned together with this list

    def _load(self):
        if self._arr is None:
            if self.frame is None or not self.frame.live:
                raise RuntimeError("keypoints were not materialised before their frame left the device")
            self._arr = self.frame.ctx.download_keypoints(self.frame.slot)
        return self._arr

    xy = property(lambda self: self._load()["xy"])
    size = property(lambda self: self._load()["size"])
    angle = property(lambda self: self._load()["angle"])
    response = property(lambda self: self._load()["response"])
    octave = property(lambda self: self._load()["octave"])

    def __len__(self):
        return self._n

    def __getitem__(self, i):
        if isinstance(i, slice):
            return [self[j] for j in range(*i.indices(len(self)))]
        if i < 0:
            i += len(self)
        if not 0 <= i < len(self):
            raise IndexError("keypoint index out of range")
        a = self._load()
        return KeyPoint(a["xy"][i, 0], a["xy"][i, 1], a["size"][i], a["angle"][i], a["response"][i], a["octave"][i])

    def __iter__(self):
        return (self[i] for i in range(len(self)))


class DeviceDescriptors:
    """The (N, 32) uint8 descriptor array of a device-resident KeyPointList; np.asarray() downloads it.
    (Holds its list weakly: frame -> nothing, list -> descriptors -> frame, so no reference cycle keeps
    a frame slot alive after the odometer has dropped the frame.)"""

    dtype = np.dtype(np.uint8)
    ndim = 2

    def __init__(self, kps):
        import weakref
        self.frame, self._n, self._kref, self._arr = kps.frame, len(kps), weakref.ref(kps), None

    shape = property(lambda self: (self._n, 32))

    def __len__(self):
        return self._n

    def _load(self):
        if self._arr is None:
            k = self._kref()
            if k is not None:
                self._arr = k._load()["desc"]
            elif self.frame is not None and self.frame.live:
                self._arr = self.frame.ctx.download_keypoints(self.frame.slot)["desc"]
            else:
                raise RuntimeError("descriptors were not materialised before their frame left the device")
        return self._arr

    def __array__(self, dtype=None, copy=None):
        a = self._load()
        return a if dtype is None else a.astype(dtype)

    def __getitem__(self, i):
        return self._load()[i]


class FrameHandle:
    """One stereo frame resident in a device slot (left image, disparity, keypoints)."""

    def __init__(self, camera, slot, w, h, roi):
        self.camera, self.ctx, self.slot = camera, camera._ctx, slot
        self.w, self.h = w, h
        self.roi = roi                      # (x0, y0, x1, y1) numpy-slice bounds, already clipped
        self._cache = {}
        self._lazy_kps = []                 # weak refs of KeyPointLists still living only in the slot

    @property
    def live(self):
        return self.slot is not None

    def crop(self, a):
        x0, y0, x1, y1 = self.roi
        return a[y0:y1, x0:x1]

    def full(self, kind):
        if kind not in self._cache:
            if not self.live:
                raise RuntimeError("frame was evicted from the device before %r was materialised" % kind)
            if kind == "disp":
                self._cache[kind] = self.ctx.download_disparity_f32(self.slot, (self.h, self.w))
            elif kind == "xyz":
                self._cache[kind] = self.ctx.download_xyz(self.slot, (self.h, self.w))
            elif kind == "left":
                self._cache[kind] = self.ctx.download_left(self.slot, (self.h, self.w))
        return self._cache[kind]

    def materialize_keypoints(self):
        """Download keypoint lists that exist only in the slot (before the slot's keypoints change)."""
        for ref in self._lazy_kps:
            k = ref()
            if k is not None:
                k._load()
        self._lazy_kps = []

    def evict(self):
        """Detach from the device slot, keeping host copies of everything."""
        if self.live:
            self.materialize_keypoints()
            for kind in ("disp", "xyz", "left"):
                self.full(kind)
            self.slot = None

    def __del__(self):
        try:
            if self.slot is not None:
                self.camera._release_slot(self.slot, self)
        except Exception:
            pass


class DeviceImage:
    """Array-like view (cropped like crop_to_valid_region_left) of a per-frame device image.

    kind: "left" (uint8 HxW), "disp" (float32 HxW) or "xyz" (float32 HxWx3).  Any numpy use
    (`np.asarray`, indexing, arithmetic) downloads the full image once and caches it."""

    _dt = {"left": np.uint8, "disp": np.float32, "xyz": np.float32, "mask": np.uint8}

    def __init__(self, frame, kind):
        self.frame, self.kind = frame, kind

    @property
    def dtype(self):
        return np.dtype(self._dt[self.kind])

    @property
    def shape(self):
        x0, y0, x1, y1 = self.frame.roi
        s = (max(y1 - y0, 0), max(x1 - x0, 0))
        return s + (3,) if self.kind == "xyz" else s

    @property
    def ndim(self):
        return len(self.shape)

    def numpy(self):
        return self.frame.crop(self.frame.full(self.kind))

    def __array__(self, dtype=None, copy=None):
        a = self.numpy()
        if dtype is not None and np.dtype(dtype) != a.dtype:
            return a.astype(dtype)
        return a.copy() if copy else a

    # The reference hands out numpy views (stereo_camera.py:53-55); user code written against it may do
    # anything an ndarray allows.  Everything below materialises the image (one download, cached) and
    # delegates: ufuncs and numpy functions called on the object, operators, and any ndarray attribute
    # or method (.astype, .reshape, .max(), .copy(), .T, .flags, ...).
    __array_priority__ = 100.0

    @staticmethod
    def _unwrap(x):
        if isinstance(x, DeviceImage):
            return x.numpy()
        if isinstance(x, (tuple, list)) and any(isinstance(v, DeviceImage) for v in x):
            return type(x)(DeviceImage._unwrap(v) for v in x)
        return x

    def __array_ufunc__(self, ufunc, method, *inputs, **kwargs):
        if "out" in kwargs:
            if any(isinstance(o, DeviceImage) for o in kwargs["out"]):
                raise TypeError("a DeviceImage is read-only: it cannot be the out= target of a ufunc")
        return getattr(ufunc, method)(*[self._unwrap(x) for x in inputs], **kwargs)

    def __array_function__(self, func, types, args, kwargs):
        return func(*[self._unwrap(a) for a in args], **{k: self._unwrap(v) for k, v in kwargs.items()})

    def __getattr__(self, name):
        # only reached for names the class does not define: forward to the materialised array
        if name.startswith("__") or name in ("frame", "kind"):
            raise AttributeError(name)
        return getattr(self.numpy(), name)

    def __getitem__(self, idx):
        return self.numpy()[idx]

    def __setitem__(self, idx, value):
        raise TypeError("a DeviceImage is read-only; use np.array(img) for a writable copy")

    def __len__(self):
        return self.shape[0]

    def __iter__(self):
        return iter(self.numpy())

    def __contains__(self, v):
        return v in self.numpy()

    def __repr__(self):
        return "DeviceImage(%s, shape=%s, dtype=%s)" % (self.kind, self.shape, self.dtype)

    def __bool__(self):
        return bool(self.numpy())

    def __float__(self):
        return float(self.numpy())

    def __int__(self):
        return int(self.numpy())

    def __copy__(self):
        return self.numpy().copy()

    def __deepcopy__(self, memo):
        return self.numpy().copy()


def _forward_operators():
    import operator as op
    binary = ("add", "sub", "mul", "truediv", "floordiv", "mod", "pow", "matmul", "and", "or", "xor", "lshift", "rshift")
    for name in binary:
        f = getattr(op, name + "_" if name in ("and", "or") else name)
        setattr(DeviceImage, "__%s__" % name, (lambda f: lambda self, o: f(self.numpy(), DeviceImage._unwrap(o)))(f))
        setattr(DeviceImage, "__r%s__" % name, (lambda f: lambda self, o: f(DeviceImage._unwrap(o), self.numpy()))(f))
    for name in ("lt", "le", "gt", "ge", "eq", "ne"):
        f = getattr(op, name)
        setattr(DeviceImage, "__%s__" % name, (lambda f: lambda self, o: f(self.numpy(), DeviceImage._unwrap(o)))(f))
    for name, f in (("neg", op.neg), ("pos", op.pos), ("abs", abs), ("invert", op.invert)):
        setattr(DeviceImage, "__%s__" % name, (lambda f: lambda self: f(self.numpy()))(f))
    DeviceImage.__hash__ = None       # like ndarray: == is elementwise, so instances are unhashable


_forward_operators()


class DisparityMask(DeviceImage):
    """feature_mask(disparity) of a device-resident disparity (stereo_odometer.py:38-41), not yet
    evaluated: ORB fuses it into its mask read; np.asarray() gives the {0,255} uint8 array."""

    def __init__(self, frame, lo, hi):
        super().__init__(frame, "mask")
        self.lo, self.hi = lo, hi

    def numpy(self):
        d = self.frame.crop(self.frame.full("disp"))
        return ((d >= self.lo) * (d <= self.hi)).astype(np.uint8) * 255


class ORB:
    """cv2.ORB_create(nfeatures) stand-in: detectAndCompute(image, mask) -> (keypoints, descriptors)."""

    def __init__(self, ctx, nfeatures=500):
        self._ctx, self.nfeatures = ctx, int(nfeatures)
        self.last_slot_args = None      # (nfeatures, mask_mode, lo16, hi16) of the latest device-side extraction

    def detectAndCompute(self, image, mask=None):
        frame = None
        if (isinstance(image, DeviceImage) and image.kind == "left" and image.frame.live and
                (mask is None or (isinstance(mask, DisparityMask) and mask.frame is image.frame))):
            # fused path: image, disparity-range mask, keypoints and descriptors never leave the GPU
            frame = image.frame
            frame.materialize_keypoints()     # an earlier list of this frame must not see the new extraction
            if mask is None:
                self._ctx.lookahead_orb(self.nfeatures, 0, 0, 0)
                self.last_slot_args = (self.nfeatures, 0, 0, 0)
                n = self._ctx.orb_slot_count(frame.slot, self.nfeatures, 0)
            else:
                # d >= lo and d <= hi on d = disp16/16 (exact in float32) <=> integer compare
                # (the reference compares float32 arrays with Python numbers: thresholds round to float32)
                lo16 = int(np.ceil(float(np.float32(mask.lo)) * 16.0))
                hi16 = int(np.floor(float(np.float32(mask.hi)) * 16.0))
                # frames prefetched from here on get their keypoints extracted right behind their SGBM
                self._ctx.lookahead_orb(self.nfeatures, 1, lo16, hi16)
                self.last_slot_args = (self.nfeatures, 1, lo16, hi16)
                n = self._ctx.orb_slot_count(frame.slot, self.nfeatures, 1, lo16, hi16)
            if n == 0:
                return (), None
            import weakref
            kps = KeyPointList(None, frame, n)
            kps.desc = DeviceDescriptors(kps)
            frame._lazy_kps += [weakref.ref(kps), weakref.ref(kps.desc)]
            return kps, kps.desc
        img = np.asarray(image)
        if img.ndim == 3:
            img = self._ctx.cvt_bgr2gray(img)
        arr = self._ctx.orb_host(img, None if mask is None else np.asarray(mask), self.nfeatures)
        kps = KeyPointList(arr, frame)
        if len(kps) == 0:
            return (), None               # cv2 returns an empty tuple and None
        kps.desc = arr["desc"]
        return kps, arr["desc"]


class BFMatcher:
    """cv2.BFMatcher.create(cv2.NORM_HAMMING) stand-in: knnMatch(query, train, k=2)."""

    def __init__(self, ctx):
        self._ctx = ctx

    def knn2_arrays(self, q, t):
        return self._ctx.bf_knn2(q, t)

    def knnMatch(self, queryDescriptors, trainDescriptors, k=2):
        if k != 2:
            raise NotImplementedError("only k=2 is implemented (what the reference uses)")
        q = np.asarray(queryDescriptors, np.uint8)
        t = np.asarray(trainDescriptors, np.uint8) if trainDescriptors is not None else np.empty((0, 32), np.uint8)
        idx, dist = self._ctx.bf_knn2(q, t)
        out = []
        for i in range(len(q)):
            out.append(tuple(DMatch(i, idx[i, j], dist[i, j]) for j in range(2) if idx[i, j] >= 0))
        return tuple(out)


class StereoSGBM:
    """cv2.StereoSGBM_create(...) stand-in: compute(left, right) -> int16 disparity x16."""

    def __init__(self, ctx, params, mode=0):
        self._ctx, self.params, self.mode = ctx, dict(params), int(params.get("mode", mode))
        ctx.set_sgbm(self.params, self.mode)

    def compute(self, left, right):
        left, right = np.asarray(left), np.asarray(right)
        if left.ndim != 2 or left.dtype != np.uint8 or left.shape != right.shape:
            raise ValueError("StereoSGBM.compute expects two uint8 single-channel images of equal size")
        return self._ctx.sgbm_compute_host(left, right)
