"""Oracle and host logic against the golden vectors generated from the REFERENCE's own numpy
code (tests/golden/make_golden.py)."""
import json
import os

import numpy as np

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def test_g1_feature_mask_host():
    from openvo_amd import StereoOdometer
    g = np.load(os.path.join(GOLD, "g1_feature_mask.npz"))
    od = StereoOdometer(None)
    m = od.feature_mask(g["disparity"])
    assert m.dtype == np.uint8 and np.array_equal(m, g["mask"])
    from oracle.odometer import RefStereoOdometer
    assert np.array_equal(RefStereoOdometer(None).feature_mask(g["disparity"]), g["mask"])


def test_g2_bilinear_oracle(oracle):
    g = np.load(os.path.join(GOLD, "g2_bilinear.npz"))
    out, st = oracle.bilinear_at(g["img"], g["xy"])
    assert np.array_equal(st == 2, g["kind"] == 2)            # ZeroDivisionError cases
    ok = g["kind"] == 0
    assert np.array_equal(np.isnan(out[ok]), np.isnan(g["out"][ok]))
    fin = ~np.isnan(g["out"][ok])
    assert np.array_equal(out[ok][fin].view(np.uint32), g["out"][ok][fin].view(np.uint32))   # bit-exact


def test_g3_rigid_body_filter_oracle(oracle):
    g3 = np.load(os.path.join(GOLD, "g3_rigid.npz"))
    n = 0
    for k in g3.files:
        if k.endswith("_mask"):
            b = k[:-5]
            m = oracle.rigid_clique(g3[b + "_prev"], g3[b + "_cur"], float(g3[b + "_thr"]))
            assert m.dtype == np.int64 and np.array_equal(m, g3[k]), b
            n += 1
    assert n == 6


def test_g4_outlier_formula():
    """The outlier pass of point_cloud_transform as the host class evaluates it."""
    g = np.load(os.path.join(GOLD, "g4_outlier.npz"))
    cur, nxt, T = g["cur"], g["nxt"], g["T"]
    h_next = np.hstack([nxt, np.ones((len(nxt), 1))]).astype(np.float64)
    h_cur = np.hstack([cur, np.ones((len(cur), 1))]).astype(np.float64)
    errors = np.linalg.norm(h_next - h_cur @ T.T, axis=1) / np.linalg.norm(h_next, axis=1)
    assert np.allclose(errors, g["errors"], rtol=1e-12, atol=1e-15)
    keep = errors < float(g["outlier_threshold"]) + np.median(errors)
    assert np.array_equal(keep, g["keep"])


class _Stereo:
    def __init__(self):
        self.i = -1

    def compute_3d(self, l, r, preprocessed=False):
        self.i += 1
        return ("3d%d" % self.i, np.zeros((2, 2), np.float32), "img%d" % self.i)


class _Orb:
    def __init__(self, stereo, script):
        self.stereo, self.script = stereo, script

    def detectAndCompute(self, img, mask):
        return ["kp"] * self.script[self.stereo.i]["n_kps"], "desc%d" % self.stereo.i


def test_g5_update_state_machine():
    """StereoOdometer.update reproduces the reference's traces decision for decision."""
    from openvo_amd import StereoOdometer
    gold = json.load(open(os.path.join(GOLD, "g5_state_machine.json")))
    assert set(gold) == {"all_ok", "few_kps_then_ok", "fallback", "double_failure", "no_prev_failure"}
    for name, case in gold.items():
        script = case["script"]
        od = StereoOdometer(None)
        st = _Stereo()
        od.stereo, od.orb = st, _Orb(st, script)
        calls = {"n": 0}

        def point_clouds(k1, k2, d1, d2, a, b):
            calls["n"] += 1
            key = "pc1" if calls["n"] == 1 else "pc2"
            if script[st.i].get(key) is None:
                return None, None
            return (key, d1, d2), (key, a, b)

        def point_cloud_transform(cp, npts):
            T = script[st.i]["T1" if cp[0] == "pc1" else "T2"]
            return None if T is None else np.array(T)

        od.point_clouds, od.point_cloud_transform = point_clouds, point_cloud_transform
        for i, exp in enumerate(case["trace"]):
            calls["n"] = 0
            ret = od.update(None, None)
            ctx = "%s step %d" % (name, i)
            assert bool(ret) == exp["ret"], ctx
            assert od.skip_cause == exp["skip_cause"], ctx
            assert od.skipped_frames == exp["skipped_frames"], ctx
            assert od.current_img == exp["current_img"] and od.prev_img == exp["prev_img"], ctx
            assert od.current_desc == exp["current_desc"], ctx
            assert getattr(od, "prev_desc", None) == exp["prev_desc"], ctx
            assert calls["n"] == exp["n_pc_calls"], ctx
            assert np.allclose(od.c_T_w, exp["c_T_w"], atol=1e-15), ctx
            assert np.allclose(od.c_T_w_prev, exp["c_T_w_prev"], atol=1e-15), ctx
            assert np.allclose(od.current_pose(), exp["pose"], atol=1e-14), ctx


def test_g6_rot2rpy():
    from openvo_amd import rot2RPY
    g = np.load(os.path.join(GOLD, "g6_rot2rpy.npz"))
    for T, exp in zip(g["T"], g["rpy"]):
        with np.errstate(all="ignore"):
            r, p, y = rot2RPY(T)
        assert r.shape == (2, 1)
        assert np.allclose(np.hstack([r, p, y]), exp, atol=1e-12, equal_nan=True)
