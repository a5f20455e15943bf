"""Host-side API parity that needs no GPU: calibration file formats and the ndarray
behaviour of compute_3d's lazy return values (the reference returns numpy views, stereo_camera.py:53-55)."""
import json
import pickle

import numpy as np
import pytest

from openvo_amd import StereoCamera, calib
from openvo_amd.features import DeviceImage, DisparityMask

SGBM = dict(minDisparity=0, numDisparities=64, blockSize=5, P1=200, P2=800, disp12MaxDiff=1, preFilterCap=63,
            uniquenessRatio=10, speckleWindowSize=100, speckleRange=2)
K_L = np.array([[498.0, 0, 325.0], [0, 502.0, 236.0], [0, 0, 1]])
K_R = np.array([[507.0, 0, 314.0], [0, 511.0, 244.0], [0, 0, 1]])
D_L = np.array([-0.11, 0.03, 0.0007, -0.0004, 0.002])
D_R = np.array([-0.09, 0.015, -0.0003, 0.0006, -0.001])
RECT = {"R": calib.rodrigues_vec_to_mat([0.012, -0.021, 0.008]), "T": np.array([-0.2, 0.004, -0.003])}


class FakeContext:
    """Records what StereoCamera.__init__ configures (the native context needs a GPU)."""

    def __init__(self):
        self.calls = {}

    def set_rectify_maps(self, cam, m1, m2):
        self.calls["maps%d" % cam] = (m1.copy(), m2.copy())

    def set_Q(self, Q):
        self.calls["Q"] = np.array(Q)

    def set_roi(self, *r):
        self.calls["roi"] = tuple(int(v) for v in r)

    def set_sgbm(self, params, mode):
        self.calls["sgbm"] = (dict(params), mode)


def _check_camera(cam, ctx):
    want = StereoCamera(K_L, D_L, K_R, D_R, RECT, SGBM, (640, 480), context=FakeContext())
    assert np.array_equal(cam.Q, want.Q) and cam.valid_region_left == want.valid_region_left
    assert cam.valid_region_right == want.valid_region_right
    for a in ("map_left_1", "map_left_2", "map_right_1", "map_right_2"):
        assert np.array_equal(getattr(cam, a), getattr(want, a))
    assert ctx.calls["sgbm"][0] == SGBM and ctx.calls["sgbm"][1] == 0
    assert ctx.calls["roi"] == tuple(cam.valid_region_left)
    assert np.array_equal(ctx.calls["Q"], cam.Q) and np.array_equal(ctx.calls["maps1"][0], cam.map_right_1)
    # crop_to_valid_region_* keep the reference's (x, y, w, h)-as-(x0, y0, x1, y1) reading [stereo_camera.py:35-41]
    img = np.arange(480 * 640).reshape(480, 640)
    vl, vr = cam.valid_region_left, cam.valid_region_right
    assert np.array_equal(cam.crop_to_valid_region_left(img), img[vl[1]:vl[3], vl[0]:vl[2]])
    assert np.array_equal(cam.crop_to_valid_region_right(img), img[vr[1]:vr[3], vr[0]:vr[2]])


def test_from_pfiles_reads_the_reference_layout(tmp_path):
    """Four pickles laid out as the reference reads them (stereo_camera.py:8-14): left/right {'K','dist'},
    rectification {'R','T'}, the ten StereoSGBM keys -- extra keys ignored, lists accepted."""
    files = [tmp_path / n for n in ("left.p", "right.p", "rect.p", "sgbm.p")]
    payload = [{"K": K_L, "dist": D_L, "rms": 0.21}, {"K": K_R.tolist(), "dist": D_R.tolist()},
               {"R": RECT["R"], "T": RECT["T"].reshape(3, 1), "E": np.eye(3)}, dict(SGBM)]
    for f, d in zip(files, payload):
        with open(f, "wb") as fh:
            pickle.dump(d, fh)
    ctx = FakeContext()
    cam = StereoCamera.from_pfiles(*files, (640, 480), context=ctx)
    _check_camera(cam, ctx)
    # positional order of the reference's signature: (left, right, rect, sgbm, img_size)
    with pytest.raises(KeyError):
        StereoCamera.from_pfiles(files[2], files[1], files[0], files[3], (640, 480), context=FakeContext())
    with pytest.raises(FileNotFoundError):
        StereoCamera.from_pfiles(tmp_path / "nope.p", files[1], files[2], files[3], (640, 480), context=FakeContext())


@pytest.mark.parametrize("ext", ["json", "npz"])
def test_from_files_safe_formats_round_trip(tmp_path, ext):
    files = [tmp_path / ("%s.%s" % (n, ext)) for n in ("left", "right", "rect", "sgbm")]
    StereoCamera.save_files(*files, K_L, D_L, K_R, D_R, RECT, SGBM)
    ctx = FakeContext()
    cam = StereoCamera.from_files(*files, (640, 480), context=ctx)
    _check_camera(cam, ctx)
    if ext == "json":
        assert set(json.load(open(files[3]))) == set(SGBM)


def test_from_files_accepts_a_rotation_vector_as_documented(tmp_path):
    """'R' of the rectification file may be the 3-number rotation vector (what cv2.stereoCalibrate users often keep):
    the camera built from it equals the one built from the 3x3 matrix; anything else is refused."""
    from openvo_amd import calib
    files = [tmp_path / ("%s.json" % n) for n in ("left", "right", "rect", "sgbm")]
    rvec = calib.rodrigues_mat_to_vec(np.asarray(RECT["R"], np.float64))
    StereoCamera.save_files(*files, K_L, D_L, K_R, D_R, {"R": rvec, "T": RECT["T"]}, SGBM)
    a = StereoCamera.from_files(*files, (640, 480), context=FakeContext())
    StereoCamera.save_files(*files, K_L, D_L, K_R, D_R, RECT, SGBM)
    b = StereoCamera.from_files(*files, (640, 480), context=FakeContext())
    assert np.allclose(a.Q, b.Q, rtol=0, atol=1e-9) and a.valid_region_left == b.valid_region_left
    assert np.abs(a.map_left_1.astype(np.int32) - b.map_left_1.astype(np.int32)).max() <= 1
    StereoCamera.save_files(*files, K_L, D_L, K_R, D_R, {"R": np.zeros(4), "T": RECT["T"]}, SGBM)
    with pytest.raises(ValueError):
        StereoCamera.from_files(*files, (640, 480), context=FakeContext())


def test_from_files_rejects_pickles_and_incomplete_files(tmp_path):
    good = [tmp_path / ("%s.json" % n) for n in ("left", "right", "rect", "sgbm")]
    StereoCamera.save_files(*good, K_L, D_L, K_R, D_R, RECT, SGBM)
    p = tmp_path / "left.p"
    with open(p, "wb") as fh:
        pickle.dump({"K": K_L, "dist": D_L}, fh)
    with pytest.raises(ValueError):
        StereoCamera.from_files(p, good[1], good[2], good[3], (640, 480), context=FakeContext())
    bad = tmp_path / "sgbm_short.json"
    json.dump({k: v for k, v in SGBM.items() if k != "P2"}, open(bad, "w"))
    with pytest.raises(KeyError):
        StereoCamera.from_files(good[0], good[1], good[2], bad, (640, 480), context=FakeContext())
    # an .npz that carries a pickled object array is refused by allow_pickle=False
    evil = tmp_path / "rect.npz"
    np.savez(evil, R=np.array([{"x": 1}], dtype=object), T=np.zeros(3))
    with pytest.raises(ValueError):
        StereoCamera.from_files(good[0], good[1], evil, good[3], (640, 480), context=FakeContext())


# ---- DeviceImage as an ndarray stand-in ------------------------------------------------------------------------
class FakeFrame:
    live = True
    slot = 0

    def __init__(self, h=12, w=16, roi=(2, 1, 14, 11)):
        rng = np.random.default_rng(0)
        self.roi = roi
        disp = rng.integers(-16, 120 * 16, (h, w)).astype(np.float32) / 16
        disp[3, 4], disp[3, 5], disp[4, 4], disp[4, 5] = 3.9375, 4.0, 100.0, 100.0625
        xyz = rng.normal(size=(h, w, 3)).astype(np.float32)
        xyz[5, 6] = np.inf
        self._full = {"disp": disp, "xyz": xyz, "left": rng.integers(0, 256, (h, w), dtype=np.uint8)}
        self.downloads = 0

    def crop(self, a):
        x0, y0, x1, y1 = self.roi
        return a[y0:y1, x0:x1]

    def full(self, kind):
        self.downloads += 1
        return self._full[kind]


def test_device_image_behaves_like_the_numpy_view_the_reference_returns():
    f = FakeFrame()
    d, x, l = DeviceImage(f, "disp"), DeviceImage(f, "xyz"), DeviceImage(f, "left")
    D, X, L = f.crop(f._full["disp"]), f.crop(f._full["xyz"]), f.crop(f._full["left"])
    assert d.shape == D.shape and x.shape == X.shape and d.dtype == np.float32 and l.dtype == np.uint8 and x.ndim == 3
    assert f.downloads == 0                                   # shape / dtype do not materialise
    # the reference's own expressions (stereo_odometer.py:38-41, 44-47, 61-79)
    m = ((d >= 4) * (d <= 100)).astype(np.uint8) * 255
    assert np.array_equal(m, ((D >= 4) * (D <= 100)).astype(np.uint8) * 255)
    assert m[2, 2] == 0 and m[2, 3] == 255 and m[3, 2] == 255 and m[3, 3] == 0      # 3.9375 / 4 / 100 / 100.0625
    assert bool(np.isinf(x[4, 4]).any()) and not np.isinf(x[0, 0]).any()
    assert np.array_equal(np.isinf(x), np.isinf(X))           # ufunc called on the object
    assert np.linalg.norm(x[int(3.7)][int(2.2)]) == np.linalg.norm(X[3][2])
    # ndarray methods / attributes / operators user code may use
    assert np.array_equal(d.astype(np.float64) / 16, D.astype(np.float64) / 16)
    assert np.array_equal((d * 16).round().astype(np.int16), (D * 16).round().astype(np.int16))
    assert d.max() == D.max() and d.min() == D.min() and d.mean() == D.mean() and d.size == D.size
    assert np.array_equal(d.reshape(-1), D.reshape(-1)) and np.array_equal(d.T, D.T) and np.array_equal(d.copy(), D)
    assert np.array_equal(-d, -D) and np.array_equal(d + d, D + D) and np.array_equal(1 - d, 1 - D) and np.array_equal(2 / (d + 200), 2 / (D + 200))
    assert np.array_equal(abs(d), abs(D)) and np.array_equal(d > l, D > L) and np.array_equal(d == D, np.ones_like(D, bool))
    assert np.array_equal(l & 15, L & 15) and np.array_equal(~l, ~L) and np.array_equal(l >> 2, L >> 2)
    assert np.array_equal(np.where(d > 50, d, 0), np.where(D > 50, D, 0))                  # __array_function__
    assert np.array_equal(np.concatenate([d, d]), np.concatenate([D, D])) and np.median(d) == np.median(D)
    assert np.array_equal(np.stack([l, l], -1), np.stack([L, L], -1)) and np.array_equal(np.asarray(x, np.float64), X.astype(np.float64))
    assert len(d) == len(D) and [r.tolist() for r in d][0] == D[0].tolist() and (100.0 in d) == (100.0 in D)
    assert np.array_equal(d[1:3, ::2], D[1:3, ::2]) and np.array_equal(x[..., 2], X[..., 2]) and d[2, 3] == D[2, 3]
    out = np.empty_like(D)
    np.add(d, 1, out=out)
    assert np.array_equal(out, D + 1)
    with pytest.raises(TypeError):
        d[0, 0] = 1.0                                          # read-only, like a view the caller must not scribble on
    with pytest.raises(TypeError):
        hash(d)
    with pytest.raises(ValueError):
        bool(d)                                                # ambiguous truth value, as for an ndarray
    w = np.array(d)
    w[0, 0] = -5
    assert d[0, 0] == D[0, 0]                                  # np.array() gives an independent copy
    # the lazy fused mask evaluates to the same array
    assert np.array_equal(np.asarray(DisparityMask(f, 4, 100)), m)
    assert "DeviceImage(disp" in repr(d)


def test_image_arguments_are_validated_before_a_pointer_reaches_native_code():
    """HxW and HxWx3 pass (HxWx1 is squeezed); anything else -- HxWx4, a scalar, mismatched shapes -- raises ValueError in
    Python.  The native side reads w * h * channels bytes from the pointer it is given (on a staging thread, where no
    exception could surface), so the check has to happen here."""
    from openvo_amd._native import _image_pair
    g = np.zeros((6, 8), np.uint8)
    l, r, ch = _image_pair(g, g.copy())
    assert ch == 1 and l.shape == (6, 8) and l.flags["C_CONTIGUOUS"]
    l, r, ch = _image_pair(g[:, :, None], g)                     # HxWx1 is the same image
    assert ch == 1 and l.shape == (6, 8)
    l, r, ch = _image_pair(np.zeros((6, 8, 3), np.uint8), np.zeros((6, 8, 3), np.int64))
    assert ch == 3 and r.dtype == np.uint8
    l, r, ch = _image_pair(np.zeros((6, 16), np.uint8)[:, ::2], g)   # non-contiguous views are copied
    assert l.flags["C_CONTIGUOUS"] and l.shape == (6, 8)
    for bad in (np.zeros((6, 8, 4), np.uint8), np.zeros((6, 8, 2), np.uint8), np.zeros(5, np.uint8), np.zeros((2, 3, 4, 3), np.uint8)):
        with pytest.raises(ValueError):
            _image_pair(bad, bad)
    with pytest.raises(ValueError):
        _image_pair(g, np.zeros((6, 9), np.uint8))
    with pytest.raises(ValueError):
        _image_pair(g, np.zeros((6, 8, 3), np.uint8))


def test_the_monocular_upload_validates_its_image_the_same_way():
    """Context.upload_mono took the channel count from ndim alone: an HxWx4 (or HxWx2) array made native code read w*h*3 bytes
    from a buffer of another size.  The check runs before the library is touched (a Context without a handle suffices)."""
    from openvo_amd._native import Context, _image

    class _Lib:
        def __init__(self):
            self.calls = []

        def vo_upload_mono(self, h, slot, ptr, w, hh, ch):
            self.calls.append((slot, w, hh, ch))
            return 0

    ctx = Context.__new__(Context)
    ctx._lib, ctx._h = _Lib(), None
    ctx._ck = lambda rc: rc
    assert ctx.upload_mono(0, np.zeros((6, 8), np.uint8)) == (8, 6)
    assert ctx.upload_mono(1, np.zeros((6, 8, 1), np.uint8)) == (8, 6)
    assert ctx.upload_mono(2, np.zeros((6, 8, 3), np.float32)) == (8, 6)
    assert ctx._lib.calls == [(0, 8, 6, 1), (1, 8, 6, 1), (2, 8, 6, 3)]
    for bad in (np.zeros((6, 8, 4), np.uint8), np.zeros((6, 8, 2), np.uint8), np.zeros(7, np.uint8), np.zeros((1, 2, 3, 3), np.uint8)):
        with pytest.raises(ValueError):
            ctx.upload_mono(0, bad)
    assert len(ctx._lib.calls) == 3
    img, ch = _image(np.zeros((6, 16, 3), np.uint8)[:, ::2])
    assert ch == 3 and img.flags["C_CONTIGUOUS"] and img.shape == (6, 8, 3)
