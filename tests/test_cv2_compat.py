"""openvo_amd.cv2_compat -- the runnable reference-side binding: a module offering exactly the
slice of cv2 the reference touches, backed by libvo355.  CPU part: with it registered as `cv2`, the reference's OWN
unmodified classes import and construct (only where /root/reference exists, i.e. in the build container).  GPU part:
the reference's per-frame call sequence (restated here, the reference's files do not travel) through the shim equals
openvo_amd's device-resident classes bit for bit."""
import os
import sys

import numpy as np
import pytest

from openvo_amd import calib
from openvo_amd import cv2_compat as shim
from openvo_amd.synth import Corridor

REF = "/root/reference/src"
SGBM = dict(minDisparity=0, numDisparities=64, blockSize=5, P1=200, P2=800, disp12MaxDiff=1, preFilterCap=63,
            uniquenessRatio=10, speckleWindowSize=100, speckleRange=2)


def test_shim_exposes_every_cv2_name_the_reference_uses():
    for name in ("stereoRectify", "initUndistortRectifyMap", "CV_16SC2", "StereoSGBM_create", "remap", "INTER_LINEAR", "cvtColor",
                 "COLOR_BGR2GRAY", "reprojectImageTo3D", "ORB_create", "BFMatcher", "NORM_HAMMING", "estimateAffine3D", "Rodrigues",
                 "putText", "FONT_HERSHEY_SIMPLEX"):
        assert hasattr(shim, name), name
    assert callable(shim.BFMatcher.create) and isinstance(shim.BFMatcher.create(shim.NORM_HAMMING), shim.BFMatcher)
    r, jac = shim.Rodrigues(calib.rodrigues_vec_to_mat([0.1, -0.2, 0.05]))
    assert np.allclose(np.ravel(r), [0.1, -0.2, 0.05], atol=1e-12) and jac is None


@pytest.mark.skipif(not os.path.isdir(REF), reason="the reference is only present in the build container")
def test_the_unmodified_reference_constructs_on_the_shim(monkeypatch):
    """sys.modules['cv2'] = shim; `from openVO import ...` then loads the reference's own files: their cv2 call sites in
    __init__ (stereoRectify, initUndistortRectifyMap, StereoSGBM_create, ORB_create, BFMatcher.create) all resolve."""
    for m in [m for m in sys.modules if m == "openVO" or m.startswith("openVO.")]:
        monkeypatch.delitem(sys.modules, m)
    monkeypatch.setitem(sys.modules, "cv2", shim)
    monkeypatch.syspath_prepend(REF)
    import openVO as ref
    assert ref.__file__.startswith(REF)
    c = Corridor("C1")
    dist = np.array([-0.08, 0.01, 0.0005, -0.0003, 0.0])
    rect = {"R": calib.rodrigues_vec_to_mat([0.002, 0.004, -0.003]), "T": np.array([-c.B, 0.001, 0.0])}
    cam = ref.StereoCamera(c.K(), dist, c.K(), dist, rect, c.sgbm_params(), (c.w, c.h))
    R1, R2, P1, P2, Q, roi1, roi2 = calib.stereo_rectify(c.K(), dist, c.K(), dist, (c.w, c.h), rect["R"], rect["T"])
    assert np.array_equal(cam.Q, Q) and tuple(cam.valid_region_left) == tuple(roi1)
    assert cam.map_left_1.shape == (c.h, c.w, 2) and cam.map_left_2.dtype == np.uint16
    odo = ref.StereoOdometer(cam, nfeatures=300, rigidity_threshold=0.1)
    assert odo.min_matches == 10 and odo.skip_cause == "" and np.array_equal(odo.current_pose(), np.eye(4))
    # numpy-only methods of the reference run on the shim-built objects
    d = np.array([[3.9375, 4.0], [100.0, 100.0625]], np.float32)
    assert odo.feature_mask(d).tolist() == [[0, 255], [255, 0]]
    assert np.array_equal(cam.crop_to_valid_region_left(np.zeros((c.h, c.w))).shape,
                          np.zeros((c.h, c.w))[roi1[1]:roi1[3], roi1[0]:roi1[2]].shape)


@pytest.mark.gpu
def test_reference_call_sequence_through_the_shim_equals_the_device_path():
    """compute_3d [stereo_camera.py:43-55] and the first half of update [stereo_odometer.py:116-117,162-175] issued
    call by call against the shim, next to openvo_amd's classes on the same pair."""
    from openvo_amd import StereoCamera, StereoOdometer
    cv2 = shim
    c = Corridor("C1")
    frames = c.pairs(0, 2)
    R1, R2, P1, P2, Q, roi, _ = cv2.stereoRectify(c.K(), c.dist(), c.K(), c.dist(), (c.w, c.h), c.rect_params()["R"], c.rect_params()["T"])
    m1, m2 = cv2.initUndistortRectifyMap(c.K(), c.dist(), R1, P1, (c.w, c.h), cv2.CV_16SC2)
    sg = cv2.StereoSGBM_create(SGBM["minDisparity"], SGBM["numDisparities"], SGBM["blockSize"], SGBM["P1"], SGBM["P2"], SGBM["disp12MaxDiff"],
                               SGBM["preFilterCap"], SGBM["uniquenessRatio"], SGBM["speckleWindowSize"], SGBM["speckleRange"])
    orb, matcher = cv2.ORB_create(nfeatures=400), cv2.BFMatcher.create(cv2.NORM_HAMMING)
    cam = StereoCamera(c.K(), c.dist(), c.K(), c.dist(), c.rect_params(), SGBM, (c.w, c.h), max_keypoints=400)
    odo = StereoOdometer(cam, nfeatures=400)
    per_frame = []
    for L, R in frames:
        bgr = np.stack([L, L, L], -1)
        gl = cv2.remap(cv2.cvtColor(bgr, cv2.COLOR_BGR2GRAY), m1, m2, cv2.INTER_LINEAR)
        gr = cv2.remap(R, m1, m2, cv2.INTER_LINEAR)
        disp = sg.compute(gl, gr).astype(np.float32) / 16
        x3 = cv2.reprojectImageTo3D(disp, Q)
        crop = lambda a: a[roi[1]:roi[3], roi[0]:roi[2]]
        mask = ((crop(disp) >= 4) * (crop(disp) <= 100)).astype(np.uint8) * 255
        kps, desc = orb.detectAndCompute(crop(gl), mask)
        per_frame.append((crop(x3), crop(disp), kps, desc))
        assert odo.update(bgr, R)
        assert np.array_equal(np.asarray(odo.current_disparity), crop(disp))
        with np.errstate(all="ignore"):
            assert np.array_equal(np.asarray(odo.current_3d).view(np.uint32), crop(x3).view(np.uint32))
        assert np.array_equal(odo.current_kps.xy, kps.xy) and np.array_equal(np.asarray(odo.current_desc), desc)
        assert isinstance(kps[0].pt, tuple)
    matches = matcher.knnMatch(per_frame[0][3], per_frame[1][3], k=2)
    good = [m[0] for m in matches if m[0].distance < 0.8 * m[1].distance]
    assert len(good) >= 10
    src = np.array([odo.bilinear_interpolate_pixels(per_frame[0][0], *per_frame[0][2][m.queryIdx].pt) for m in good])
    dst = np.array([odo.bilinear_interpolate_pixels(per_frame[1][0], *per_frame[1][2][m.trainIdx].pt) for m in good])
    T, scale = cv2.estimateAffine3D(src, dst, force_rotation=True)
    assert np.allclose(np.vstack([T, [0, 0, 0, 1]]), odo.c_T_w, rtol=0, atol=1e-9)      # default thresholds: one plain fit
    rot, _ = cv2.Rodrigues(T[:, :3])
    assert rot.shape == (3, 1) and np.linalg.norm(rot) < 0.1
