"""Independent evidence for the oracle (CPU only).  Every cv2 stage of the oracle is unpinned -- the reference holds
no fixtures and OpenCV is not installable here -- so where a third party on this box implements the same
definition, the oracle is checked against it: SciPy for Rodrigues, the rigid fit, the 3x3 median and Hamming kNN;
scikit-image's copy of the rBRIEF sampling table.  The oracle's known-answer tests also run once under
AddressSanitizer + UndefinedBehaviorSanitizer (oracle/Makefile `asan`).  This does not replace
tests/test_cv2_crosscheck.py (which needs a real cv2); it replaces "a review" with checks."""
import hashlib
import os
import re
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def O():
    from oracle import oracle
    oracle.build_oracle()
    return oracle


def test_rodrigues_against_scipy_rotvec(O):
    """cv2.Rodrigues(R) (stereo_odometer.py:212) = the rotation vector of R: SciPy's Rotation.as_rotvec."""
    from scipy.spatial.transform import Rotation
    rng = np.random.default_rng(3)
    vecs = list(rng.normal(scale=0.6, size=(40, 3))) + [np.zeros(3), np.array([1e-9, 0, 0]), np.array([0, np.pi - 1e-3, 0]),
                                                      np.array([0.7, -0.7, 0.1]) * 3.0 / np.linalg.norm([0.7, -0.7, 0.1])]
    for v in vecs:
        R = Rotation.from_rotvec(v).as_matrix()
        got = O.rodrigues(R).reshape(3)
        want = Rotation.from_matrix(R).as_rotvec()
        assert np.allclose(got, want, rtol=0, atol=1e-9), (v, got, want)


def test_umeyama_against_scipy_align_vectors_on_exact_motions(O):
    """estimateAffine3D(src, dst, force_rotation=True) (stereo_odometer.py:190,204) on exact rigid motions: the rotation is
    SciPy's Kabsch solution (Rotation.align_vectors on the centred clouds), the translation follows, scale 1."""
    from scipy.spatial.transform import Rotation
    rng = np.random.default_rng(5)
    for n in (3, 4, 12, 200):
        src = rng.uniform(-3, 3, (n, 3)).astype(np.float32)
        Rt = Rotation.from_rotvec(rng.normal(scale=0.4, size=3))
        t = rng.uniform(-1, 1, 3)
        dst = (Rt.apply(src.astype(np.float64)) + t).astype(np.float32)
        T, s = O.umeyama(src, dst, True)
        a, b = src.astype(np.float64), dst.astype(np.float64)
        Rs, _ = Rotation.align_vectors(b - b.mean(0), a - a.mean(0))     # rotation taking src onto dst
        assert np.allclose(T[:, :3] / s, Rs.as_matrix(), rtol=0, atol=2e-5), n
        assert abs(s - 1.0) < 1e-4 and np.allclose(T[:, 3], b.mean(0) - T[:, :3] @ a.mean(0), rtol=0, atol=1e-6)
        assert np.allclose(T[:, :3] @ a.T + T[:, 3:4], b.T, rtol=0, atol=5e-4)
    # a reflection is not a rotation: force_rotation must return det +1
    src = rng.uniform(-1, 1, (30, 3)).astype(np.float32)
    dst = src * np.array([1, 1, -1], np.float32)
    T, s = O.umeyama(src, dst, True)
    assert np.linalg.det(T[:, :3] / s) > 0.999


def test_median3x3_against_scipy_ndimage(O):
    """medianBlur(3) on int16 with a replicated border (StereoSGBM's post filter) = ndimage.median_filter(mode='nearest')."""
    from scipy import ndimage
    rng = np.random.default_rng(7)
    for shape in ((5, 7), (64, 48), (3, 3), (1, 9), (17, 1)):
        img = rng.integers(-16, 2048, shape).astype(np.int16)
        img[rng.random(shape) < 0.2] = -16
        assert np.array_equal(O.median3x3_s16(img), ndimage.median_filter(img, size=3, mode="nearest")), shape


def test_hamming_knn2_with_ties_against_scipy_cdist(O):
    """BFMatcher(NORM_HAMMING).knnMatch(k=2) (stereo_odometer.py:163): the two smallest Hamming distances per query and,
    on ties, the lower train index first -- against a full cdist table sorted stably."""
    from scipy.spatial.distance import cdist
    rng = np.random.default_rng(11)
    q = rng.integers(0, 256, (70, 32), dtype=np.uint8)
    t = rng.integers(0, 256, (90, 32), dtype=np.uint8)
    t[10] = t[3]; t[55] = t[3]; q[0] = t[3]                       # exact ties at distance 0
    t[20, :31] = t[21, :31]; t[20, 31] ^= 1                       # near duplicates
    D = np.rint(cdist(np.unpackbits(q, axis=1), np.unpackbits(t, axis=1), "hamming") * 256).astype(np.int32)
    idx, dist = O.bf_knn2_hamming(q, t)
    order = np.argsort(D, axis=1, kind="stable")[:, :2]
    assert np.array_equal(idx, order.astype(np.int32))
    assert np.array_equal(dist, np.take_along_axis(D, order, 1))
    assert idx[0, 0] == 3 and idx[0, 1] == 10 and dist[0].tolist() == [0, 0]


def _pattern():
    txt = open(os.path.join(ROOT, "include", "vo_orb_pattern.inc")).read()
    body = re.sub(r"//.*", "", re.sub(r"/\*.*?\*/", "", txt, flags=re.S))
    return np.array([int(x) for x in re.findall(r"-?\d+", body)], np.int8)


def test_brief_sampling_table_is_the_published_one():
    """include/vo_orb_pattern.inc is shared by the HIP product and the oracle, so no HIP-vs-oracle test can see a
    transcription slip in it: its digest is pinned here, and where scikit-image's copy of the same published table
    (orb_descriptor_positions.txt, Rublee et al. 2011) is on the box it is compared entry by entry."""
    a = _pattern()
    assert a.shape == (1024,) and a[:8].tolist() == [8, -3, 9, 5, 4, 2, 7, -12] and a[-4:].tolist() == [-1, -6, 0, -11]
    assert np.abs(a).max() <= 15                                  # inside the 31 x 31 patch
    assert hashlib.sha256(a.tobytes()).hexdigest() == "2164181aea6ff9ac426ca512d5130d15e1f6e3cd47b1cbdd568bbe1e55d49023"
    import glob
    for path in glob.glob("/opt/conda/lib/python3*/site-packages/skimage/feature/orb_descriptor_positions.txt") + \
            glob.glob("/usr/lib/python3*/site-packages/skimage/feature/orb_descriptor_positions.txt"):
        sk = np.loadtxt(path).astype(np.int8)
        assert sk.shape == (256, 4) and np.array_equal(sk.reshape(-1), a)


def test_oracle_known_answers_under_address_and_ub_sanitizers():
    """The yardstick every parity test leans on, once under ASan + UBSan: the oracle's known-answer tests in a child
    interpreter with libasan preloaded and oracle/libvo_oracle_asan.so loaded instead of the -O2 build."""
    asan = subprocess.run(["gcc", "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    if not asan or not os.path.exists(asan):
        pytest.skip("libasan is not installed")
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "asan"], stdout=subprocess.DEVNULL)
    lib = os.path.join(ROOT, "oracle", "libvo_oracle_asan.so")
    env = dict(os.environ, LD_PRELOAD=asan, ASAN_OPTIONS="detect_leaks=0:abort_on_error=1", UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1",
               VO_ORACLE_LIB=lib, PYTHONPATH=ROOT)
    r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-p", "no:cacheprovider", os.path.join(ROOT, "tests", "test_oracle_known_answers.py"),
                        os.path.join(ROOT, "tests", "test_golden.py")], cwd=ROOT, env=env, capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert "passed" in r.stdout and "ERROR: AddressSanitizer" not in r.stderr and "runtime error" not in r.stderr
