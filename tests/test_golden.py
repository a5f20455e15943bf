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


# ---- G7 / G8: every decision of point_clouds / point_cloud_transform, pinned by the reference run with scripted cv2 stand-ins ----
def _g7_cases():
    g = np.load(os.path.join(GOLD, "g7_point_clouds.npz"))
    names = sorted({k.split("__")[0] for k in g.files if "__" in k})
    assert names == ["edges_default", "edges_thr07", "min_matches_5", "min_matches_5_short", "nine_survivors", "no_matches",
                     "ten_survivors"]
    return g, names


def _g7_table(g, name):
    """(idx, dist) in the layout of a kNN-2 result: row i = query i, columns = best, second"""
    rows, train, second = g[name + "__rows"].reshape(-1, 2), g[name + "__train"], g[name + "__second"]
    idx = np.stack([train, second], 1).astype(np.int32).reshape(-1, 2)
    return idx, rows.astype(np.int32)


def test_g7_ratio_test_native_and_oracle(oracle):
    """The strict `<` of the ratio test, on float32 distances against a double product (stereo_odometer.py:164): the
    library's vo_ratio_filter and the oracle's keep exactly the matches the reference kept, in its order."""
    import ctypes
    from openvo_amd import _native
    g, names = _g7_cases()
    L = _native.lib()
    n_checked = 0
    for name in names:
        idx, dist = _g7_table(g, name)
        thr = float(g[name + "__match_threshold"])
        if len(idx) == 0:
            continue
        want_q, want_t = g[name + "__q_idx"], g[name + "__t_idx"]
        if bool(g[name + "__none"]):
            want_n = None                      # the reference returned before sampling: only the count is observable (< min_matches)
        else:
            want_n = len(want_q)
        q = np.zeros(len(idx), np.int32); t = np.zeros(len(idx), np.int32); m = ctypes.c_int(0)
        assert L.vo_ratio_filter(idx.ctypes.data_as(ctypes.c_void_p), dist.ctypes.data_as(ctypes.c_void_p), len(idx), ctypes.c_double(thr),
                                 q.ctypes.data_as(ctypes.c_void_p), t.ctypes.data_as(ctypes.c_void_p), ctypes.byref(m)) == 0
        oq, ot = oracle.ratio_filter(idx, dist, thr)
        assert np.array_equal(oq, q[:m.value]) and np.array_equal(ot, t[:m.value]), name
        if want_n is None:
            assert m.value < int(g[name + "__min_matches"]), name
        else:
            assert m.value == want_n and np.array_equal(q[:m.value], want_q) and np.array_equal(t[:m.value], want_t), name
            n_checked += want_n
    assert n_checked > 80


def test_g7_point_clouds_oracle(oracle, monkeypatch):
    """oracle/odometer.py's point_clouds on the scripted match tables: same None / arrays, same query -> frame 1 and
    train -> frame 2 mapping, 3-D points bit-exact (stereo_odometer.py:162-175)."""
    from oracle import oracle as O
    from oracle.odometer import RefStereoOdometer
    g, names = _g7_cases()
    im1, im2 = g["im1"], g["im2"]

    class Dense:
        def __init__(self, a):
            self.a = a

        def sample(self, xy):
            return O.bilinear_at(self.a, xy)

    for name in names:
        idx, dist = _g7_table(g, name)
        monkeypatch.setattr(O, "bf_knn2_hamming", lambda a, b, _i=idx, _d=dist: (_i, _d))
        od = RefStereoOdometer(None, match_threshold=float(g[name + "__match_threshold"]), min_matches=int(g[name + "__min_matches"]))
        f1 = dict(desc="desc1", d3=Dense(im1), kps=dict(xy=g[name + "__kp1"].reshape(-1, 2)))
        f2 = dict(desc="desc2", d3=Dense(im2), kps=dict(xy=g[name + "__kp2"].reshape(-1, 2)))
        p1, p2 = od.point_clouds(f1, f2)
        if bool(g[name + "__none"]):
            assert p1 is None and p2 is None, name
        else:
            assert np.array_equal(od.last_matches[0], g[name + "__q_idx"]) and np.array_equal(od.last_matches[1], g[name + "__t_idx"]), name
            for got, want in ((p1, g[name + "__pts1"]), (p2, g[name + "__pts2"])):
                assert got.dtype == np.float32 and np.array_equal(got.view(np.uint32), want.view(np.uint32)), name


def _g8_cases():
    g = np.load(os.path.join(GOLD, "g8_point_cloud_transform.npz"))
    names = [str(n) for n in g["names"]]
    assert len(names) == 21 and {str(g[n + "__skip_cause"]) for n in names} == {"init", "rigidity", "outlier", "nan", "bigdist", "bigrot"}
    return g, names


def _g8_expect(g, n, ret, skip_cause, ctx):
    assert (ret is None) == bool(g[n + "__ret_none"]), ctx
    assert skip_cause == str(g[n + "__skip_cause"]), ctx
    if ret is not None:
        assert np.array_equal(np.asarray(ret), g[n + "__ret"]), ctx


class _CtxStub:
    def __init__(self, rvec):
        self.rvec, self.n_rodrigues = np.asarray(rvec, np.float64), 0

    def rodrigues(self, R):
        self.n_rodrigues += 1
        return self.rvec.copy()


def test_g8_point_cloud_transform_host_generic_path():
    """openvo_amd.StereoOdometer.point_cloud_transform through its public seams (estimate / rigid filter / Rodrigues handed
    back from the fixture): rigidity < 10 vs "outlier" precedence, the quirk that "rigidity" is set while a T is still
    returned when min_matches < 10, nan, both gates at skipped_frames 0 / 1 / 2, "bigrot" written after "bigdist", the
    strict `>` of both gates -- decision for decision what the reference did (stereo_odometer.py:177-223)."""
    from openvo_amd import StereoOdometer
    g, names = _g8_cases()
    for n in names:
        rig, out, mm, skipped = g[n + "__kw"]
        od = StereoOdometer(None, rigidity_threshold=float(rig), outlier_threshold=float(out), min_matches=int(mm))
        od.skipped_frames, od.skip_cause = int(skipped), "init"
        od._ctx = _CtxStub(g[n + "__rvec"])
        queue, lens = [T for T in g[n + "__Ts"]], []

        def estimate(src, dst, _q=queue, _l=lens):
            _l.append(len(src))
            return np.vstack([_q.pop(0), [0, 0, 0, 1]])

        od._estimate = estimate
        od.rigid_body_filter = lambda a, b, _m=g[n + "__mask"]: _m
        with np.errstate(all="ignore"):
            ret = od.point_cloud_transform(g[n + "__prev"].copy(), g[n + "__cur"].copy())
        _g8_expect(g, n, ret, od.skip_cause, n)
        assert lens == g[n + "__est_lens"].tolist() and not queue, n              # same fits on the same number of points
        assert od._ctx.n_rodrigues == int(g[n + "__n_rodrigues"]), n


def test_g8_fused_path_decision_table():
    """_pair_fused turns the native step's record (M, n1, n2, flags, fit statuses, T) into the reference's decisions: fed
    with the counts the reference run produced, it returns the same T / None and leaves the same skip_cause."""
    from openvo_amd import StereoOdometer
    g, names = _g8_cases()

    class Stereo:
        def slot_key(self, s):
            return (s, 0)

    for n in names:
        rig, out, mm, skipped = g[n + "__kw"]
        mm = int(mm)
        lens = g[n + "__est_lens"].tolist()
        M = len(g[n + "__prev"])
        n1 = int(g[n + "__mask"].sum()) if rig > 0 else M
        ran_outlier = out > 0 and n1 >= 10
        final_fit = len(lens) == (2 if ran_outlier else 1)
        n2 = lens[-1] if final_fit else (mm - 1 if ran_outlier else n1)
        Ts = g[n + "__Ts"]
        od = StereoOdometer(None, rigidity_threshold=float(rig), outlier_threshold=float(out), min_matches=mm)
        od.stereo, od.skipped_frames, od.skip_cause = Stereo(), int(skipped), "init"
        stub = _CtxStub(g[n + "__rvec"])
        rec = (np.array([M, n1, n2, 0], np.int32), np.array([0 if ran_outlier else 1, 0 if final_fit else 1], np.int32), None,
               Ts[-1] if final_fit else np.zeros((3, 4)))
        stub.pose_pair = lambda a, b, *params, _r=rec: _r
        od._ctx = stub
        if M < mm:
            continue                                   # (point_clouds would have returned None first: not a case of this table)
        with np.errstate(all="ignore"):
            ret = od._pair_fused(0, 1)
        _g8_expect(g, n, ret, od.skip_cause, n)
        assert stub.n_rodrigues == int(g[n + "__n_rodrigues"]), n


def test_g8_point_cloud_transform_oracle(monkeypatch):
    """oracle/odometer.py's restatement of point_cloud_transform, same fixture (its cv2 replacements scripted)."""
    from oracle import oracle as O
    from oracle.odometer import RefStereoOdometer
    g, names = _g8_cases()
    for n in names:
        rig, out, mm, skipped = g[n + "__kw"]
        od = RefStereoOdometer(None, rigidity_threshold=float(rig), outlier_threshold=float(out), min_matches=int(mm))
        od.skipped_frames, od.skip_cause = int(skipped), "init"
        queue, lens = [T for T in g[n + "__Ts"]], []
        monkeypatch.setattr(O, "umeyama", lambda s, d, fr=True, _q=queue, _l=lens: (_l.append(len(s)) or _q.pop(0), 1.0))
        monkeypatch.setattr(O, "rodrigues", lambda R, _v=g[n + "__rvec"]: _v.copy())
        monkeypatch.setattr(O, "rigid_clique", lambda a, b, thr, _m=g[n + "__mask"]: _m)
        with np.errstate(all="ignore"):
            ret = od.point_cloud_transform(g[n + "__prev"].copy(), g[n + "__cur"].copy())
        _g8_expect(g, n, ret, od.skip_cause, n)
        assert lens == g[n + "__est_lens"].tolist() and not queue, n
