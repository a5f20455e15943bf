"""Generate golden vectors from the REFERENCE's own numpy-only code (G7 / G8 added in
round 4: every decision point_clouds and point_cloud_transform make, with scripted stand-ins where they call cv2).

Run in the build container only (needs /root/reference):  python tests/golden/make_golden.py
The reference's heavy arithmetic lives in cv2, which is absent here; the functions exercised
below are the reference's pure-numpy ones (feature_mask, bilinear_interpolate_pixels,
rigid_body_filter, the outlier formula of point_cloud_transform, the update() state machine,
rot2RPY).  A stand-in module object is registered as sys.modules["cv2"] only so that
`import openVO` succeeds; it carries no arithmetic.  Outputs: small .npz / .json fixtures
next to this script (data only -- inputs and expected outputs).
"""
import json
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))

cv2 = types.ModuleType("cv2")
cv2.NORM_HAMMING = 6
cv2.ORB_create = lambda nfeatures=500: None
cv2.BFMatcher = types.SimpleNamespace(create=lambda norm: None)
sys.modules["cv2"] = cv2
sys.path.insert(0, "/root/reference/src")
from openVO import StereoOdometer, rot2RPY  # noqa: E402


def new_odom(**kw):
    return StereoOdometer(stereo_camera=None, **kw)


def g1_feature_mask():
    rng = np.random.default_rng(101)
    d = (rng.integers(-16, 130 * 16, size=(37, 53)) / 16.0).astype(np.float32)
    d[0, :4] = [3.9375, 4.0, 100.0, 100.0625]
    d[1, :3] = [-1.0, 0.0, np.inf]
    out = new_odom().feature_mask(d)
    np.savez(os.path.join(HERE, "g1_feature_mask.npz"), disparity=d, mask=out)


def g2_bilinear():
    rng = np.random.default_rng(202)
    h, w = 9, 11
    img = rng.normal(size=(h, w, 3)).astype(np.float32) * 10
    img[2, 3] = np.inf          # one inf neighbour
    img[5, 5] = [1.0, -np.inf, 2.0]
    img[5, 6] = np.inf          # two inf neighbours in a cell
    img[6, 5] = np.inf          # three
    img[7, 8] = np.nan          # NaN is NOT excluded
    img[0, 0] = np.inf
    img[0, 1] = np.inf
    img[1, 0] = np.inf
    img[1, 1] = np.inf          # all four inf around (0.5, 0.5)
    od = new_odom()
    pts, outs, kinds = [], [], []
    cases = [(4.25, 3.5), (1.75, 2.25), (2.5, 1.5), (3.0, 2.0), (5.5, 5.5), (5.25, 4.75), (4.5, 5.5),
             (10.0, 4.5), (10.5, 4.5), (3.5, 8.0), (3.5, 8.75), (10.25, 8.5), (7.5, 6.5), (8.0, 7.0),
             (0.5, 0.5), (2.0, 3.0), (3.0, 2.0), (2.0, 2.0), (9.999, 7.999), (6.0, 0.0)]
    for i in range(40):
        cases.append((float(np.float32(rng.uniform(0, w - 0.01))), float(np.float32(rng.uniform(0, h - 0.01)))))
    for (x, y) in cases:
        x, y = float(np.float32(x)), float(np.float32(y))
        try:
            with np.errstate(all="ignore"):
                r = od.bilinear_interpolate_pixels(img, x, y)
            kind = 0
            r = np.asarray(r)
            assert r.dtype == np.float32, r.dtype
        except ZeroDivisionError:
            kind, r = 2, np.full(3, np.nan, np.float32)
        pts.append((x, y)); outs.append(r); kinds.append(kind)
    np.savez(os.path.join(HERE, "g2_bilinear.npz"), img=img, xy=np.array(pts, np.float32),
             out=np.array(outs, np.float32), kind=np.array(kinds, np.uint8))


def g3_rigid():
    rng = np.random.default_rng(303)
    out = {}
    for m, thr, nout in [(12, 0.05, 3), (64, 0.1, 10), (64, 0.02, 10), (300, 0.1, 40), (300, 0.7, 40), (5, 0.1, 0)]:
        prev = (rng.uniform(-5, 5, size=(m, 3)) * [1, 0.3, 4] + [0, 0, 12]).astype(np.float32)
        ang = 0.02
        Rm = np.array([[np.cos(ang), 0, np.sin(ang)], [0, 1, 0], [-np.sin(ang), 0, np.cos(ang)]])
        cur = (prev @ Rm.T + [0.01, 0.0, -0.25] + rng.normal(scale=0.004, size=(m, 3))).astype(np.float32)
        bad = rng.choice(m, nout, replace=False)
        cur[bad] += rng.normal(scale=2.0, size=(nout, 3)).astype(np.float32)
        od = new_odom(rigidity_threshold=thr)
        mask = od.rigid_body_filter(prev, cur)
        key = "m%d_t%g" % (m, thr)
        out[key + "_prev"], out[key + "_cur"], out[key + "_mask"] = prev, cur, np.asarray(mask)
        out[key + "_thr"] = np.float64(thr)
    np.savez(os.path.join(HERE, "g3_rigid.npz"), **out)


def g4_outlier():
    """The outlier pass of point_cloud_transform (reference stereo_odometer.py:191-197) with a
    supplied T (cv2.estimateAffine3D is not available); formula restated from those lines."""
    rng = np.random.default_rng(404)
    m = 50
    cur = (rng.uniform(-5, 5, size=(m, 3)) + [0, 0, 10]).astype(np.float32)
    T = np.eye(4)
    T[:3, 3] = [0.02, -0.01, -0.25]
    nxt = (cur @ T[:3, :3].T + T[:3, 3] + rng.normal(scale=0.01, size=(m, 3))).astype(np.float32)
    nxt[::7] += 1.0
    outlier_threshold = 0.01
    h_pts = np.hstack([nxt, np.array([[1] * len(nxt)]).transpose()])
    h_prev = np.hstack([cur, np.array([[1] * len(cur)]).transpose()])
    errors = np.array([np.linalg.norm(h_pts[i] - T @ h_prev[i]) / np.linalg.norm(h_pts[i]) for i in range(len(h_pts))])
    threshold = outlier_threshold + np.median(errors)
    keep = errors < threshold
    np.savez(os.path.join(HERE, "g4_outlier.npz"), cur=cur, nxt=nxt, T=T, errors=errors,
             threshold=np.float64(threshold), keep=keep, outlier_threshold=np.float64(outlier_threshold))


class _Stereo:
    def __init__(self, script):
        self.i = -1
        self.script = script

    def compute_3d(self, l, r, preprocessed=False):
        self.i += 1
        return ("3d%d" % self.i, np.zeros((2, 2), np.float32), "img%d" % self.i)


class _Orb:
    def __init__(self, stereo, script):
        self.stereo, self.script = stereo, script

    def detectAndCompute(self, img, mask):
        n = self.script[self.stereo.i]["n_kps"]
        return ["kp"] * n, "desc%d" % self.stereo.i


def g5_state_machine():
    def Tz(dz, ang=0.0):
        T = np.eye(4)
        c, s = np.cos(ang), np.sin(ang)
        T[:3, :3] = [[c, 0, s], [0, 1, 0], [-s, 0, c]]
        T[:3, 3] = [0.01, 0, dz]
        return T
    # each step: n_kps; pc1 = result of first point_clouds call ("ok"/None); T1 = first transform;
    # pc2/T2 = fallback attempt against prev frame
    scripts = {
        "all_ok": [dict(n_kps=50)] + [dict(n_kps=50, pc1="ok", T1=Tz(-0.25, 0.01 * i)) for i in range(4)],
        "few_kps_then_ok": [dict(n_kps=3), dict(n_kps=50), dict(n_kps=5), dict(n_kps=50, pc1="ok", T1=Tz(-0.5))],
        "fallback": [dict(n_kps=50), dict(n_kps=50, pc1="ok", T1=Tz(-0.25)),
                     dict(n_kps=50, pc1="ok", T1=None, pc2="ok", T2=Tz(-0.5)),
                     dict(n_kps=50, pc1=None, pc2="ok", T2=Tz(-0.3)),
                     dict(n_kps=50, pc1="ok", T1=Tz(-0.2))],
        "double_failure": [dict(n_kps=50), dict(n_kps=50, pc1="ok", T1=Tz(-0.25)),
                           dict(n_kps=50, pc1="ok", T1=None, pc2="ok", T2=None),
                           dict(n_kps=50, pc1=None, pc2=None),
                           dict(n_kps=50, pc1="ok", T1=Tz(-0.7)),
                           dict(n_kps=50, pc1="ok", T1=None, pc2=None)],
        "no_prev_failure": [dict(n_kps=50), dict(n_kps=50, pc1="ok", T1=None), dict(n_kps=50, pc1=None),
                            dict(n_kps=50, pc1="ok", T1=Tz(-0.2))],
    }
    out = {}
    for name, script in scripts.items():
        od = new_odom()
        st = _Stereo(script)
        od.stereo = st
        od.orb = _Orb(st, script)
        calls = {"n": 0}

        def point_clouds(k1, k2, d1, d2, a, b, _s=script, _st=st, _c=calls):
            _c["n"] += 1
            step = _s[_st.i]
            key = "pc1" if _c["n"] == 1 else "pc2"
            if step.get(key) is None:
                return None, None
            return (key, d1, d2), (key, a, b)

        def point_cloud_transform(cp, npts, _s=script, _st=st):
            step = _s[_st.i]
            return step["T1"] if cp[0] == "pc1" else step["T2"]

        od.point_clouds = point_clouds
        od.point_cloud_transform = point_cloud_transform
        trace = []
        for i in range(len(script)):
            calls["n"] = 0
            ret = od.update(None, None)
            trace.append(dict(ret=bool(ret), skip_cause=od.skip_cause, skipped_frames=int(od.skipped_frames),
                              current_img=od.current_img, prev_img=od.prev_img,
                              current_desc=od.current_desc, prev_desc=getattr(od, "prev_desc", None),
                              n_pc_calls=calls["n"], c_T_w=od.c_T_w.tolist(), c_T_w_prev=od.c_T_w_prev.tolist(),
                              pose=od.current_pose().tolist()))
        ser = [{k: (None if v is None else (v.tolist() if isinstance(v, np.ndarray) else v)) for k, v in s.items()}
               for s in script]
        out[name] = dict(script=ser, trace=trace)
    json.dump(out, open(os.path.join(HERE, "g5_state_machine.json"), "w"), indent=1)


def g6_rot2rpy():
    rng = np.random.default_rng(606)
    Ts, outs = [], []
    for i in range(12):
        a = rng.uniform(-np.pi, np.pi, 3)
        if i == 10:
            a[1] = np.pi / 2
        if i == 11:
            a[1] = -np.pi / 2
        cx, sx, cy, sy, cz, sz = np.cos(a[0]), np.sin(a[0]), np.cos(a[1]), np.sin(a[1]), np.cos(a[2]), np.sin(a[2])
        R = np.array([[cz, -sz, 0], [sz, cz, 0], [0, 0, 1]]) @ np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]]) @ \
            np.array([[1, 0, 0], [0, cx, -sx], [0, sx, cx]])
        T = np.eye(4)
        T[:3, :3] = R
        with np.errstate(all="ignore"):
            r, p, y = rot2RPY(T)
        Ts.append(T); outs.append(np.hstack([r, p, y]))
    np.savez(os.path.join(HERE, "g6_rot2rpy.npz"), T=np.array(Ts), rpy=np.array(outs))


# ---- G7: point_clouds (reference stereo_odometer.py:162-175) with a scripted matcher -----------------------------------------
class _KP:
    def __init__(self, x, y):
        self.pt = (x, y)


class _DM:
    def __init__(self, q, t, d):
        self.queryIdx, self.trainIdx, self.distance = int(q), int(t), float(d)


class _Matcher:
    """matcher.knnMatch(desc1, desc2, k=2) stand-in: hands back a prepared table (control flow only, no arithmetic)."""

    def __init__(self, table):
        self.table, self.calls = table, []

    def knnMatch(self, d1, d2, k=2):
        self.calls.append((d1, d2, k))
        return self.table


def g7_point_clouds():
    rng = np.random.default_rng(707)
    h, w = 40, 56
    im1 = (rng.normal(size=(h, w, 3)) * 5 + [0, 0, 12]).astype(np.float32)
    im2 = (rng.normal(size=(h, w, 3)) * 5 + [0, 0, 12]).astype(np.float32)
    im1[7, 9] = np.inf                       # an inf tap next to a keypoint: excluded from the weights, not an error
    im2[20, 30] = [np.inf, 1.0, 2.0]
    out = dict(im1=im1, im2=im2)
    cases = {}

    def table_edges():
        # (best, second) distance pairs at and around best == ratio * second, for every second distance 5 .. 255
        rows = []
        for d1 in range(5, 256, 5):
            for off in (-1, 0, 1):
                rows.append((round(0.8 * d1) + off, d1))
        rows += [(0, 0), (0, 1), (1, 1), (255, 256), (204, 255), (205, 256), (3, 4), (2, 3)]
        return rows

    def run(name, rows, nq_extra=0, **kw):
        n = len(rows)
        kp1 = [(float(np.float32(rng.uniform(0, w - 1.001))), float(np.float32(rng.uniform(0, h - 1.001)))) for _ in range(n)]
        ntrain = n + 7
        kp2 = [(float(np.float32(rng.uniform(0, w - 1.001))), float(np.float32(rng.uniform(0, h - 1.001)))) for _ in range(ntrain)]
        if n > 3:
            kp1[0] = (8.5, 6.5)              # the cell with the inf tap of im1
            kp1[3] = (55.0, 39.0)            # the last pixel: no right / lower neighbour
        train = rng.permutation(ntrain)[:n]
        second = (train + 1 + rng.integers(0, ntrain - 1, n)) % ntrain
        table = [(_DM(i, train[i], rows[i][0]), _DM(i, second[i], rows[i][1])) for i in range(n)]
        od = new_odom(**kw)
        od.matcher = _Matcher(table)
        real_bilinear, seen = od.bilinear_interpolate_pixels, []

        def bilinear(img, x, y, _f=real_bilinear, _s=seen):          # which image was sampled where, in call order
            _s.append((1 if img is im1 else 2, x, y))
            return _f(img, x, y)

        od.bilinear_interpolate_pixels = bilinear
        with np.errstate(all="ignore"):
            p1, p2 = od.point_clouds([_KP(*p) for p in kp1], [_KP(*p) for p in kp2], "desc1", "desc2", im1, im2)
        assert od.matcher.calls == [("desc1", "desc2", 2)]
        # the surviving matches, recovered from the coordinates the reference sampled (every keypoint position is unique)
        q_idx = [kp1.index((x, y)) for (im, x, y) in seen if im == 1]
        t_idx = [kp2.index((x, y)) for (im, x, y) in seen if im == 2]
        assert [im for (im, _, _) in seen] == [1, 2] * len(q_idx)
        c = dict(q_idx=np.array(q_idx, np.int64), t_idx=np.array(t_idx, np.int64),rows=np.array(rows, np.float64), train=train.astype(np.int64), second=second.astype(np.int64),
                 kp1=np.array(kp1, np.float32), kp2=np.array(kp2, np.float32), none=np.array(p1 is None),
                 match_threshold=np.float64(od.match_threshold), min_matches=np.int64(od.min_matches))
        if p1 is not None:
            assert p1.dtype == np.float32 and p2.dtype == np.float32, (p1.dtype, p2.dtype)
            c["pts1"], c["pts2"] = p1, p2
        cases[name] = c

    run("edges_default", table_edges())
    run("edges_thr07", [(round(0.7 * d1) + off, d1) for d1 in range(10, 200, 10) for off in (-1, 0, 1)], match_threshold=0.7)
    rows9 = [(10, 50)] * 9 + [(40, 50)] * 6          # nine survivors of fifteen: one short of min_matches
    run("nine_survivors", rows9)
    run("ten_survivors", rows9 + [(39, 50)])
    run("min_matches_5", [(10, 50)] * 5 + [(45, 50)] * 3, min_matches=5)
    run("min_matches_5_short", [(10, 50)] * 4 + [(45, 50)] * 3, min_matches=5)
    run("no_matches", [])
    for name, c in cases.items():
        for k, v in c.items():
            out["%s__%s" % (name, k)] = v
    np.savez(os.path.join(HERE, "g7_point_clouds.npz"), **out)


# ---- G8: point_cloud_transform (reference stereo_odometer.py:177-223) with scripted cv2.estimateAffine3D / cv2.Rodrigues -------
def g8_point_cloud_transform():
    rng = np.random.default_rng(808)

    def T34(t, ang=0.0):
        c, s = np.cos(ang), np.sin(ang)
        return np.array([[c, 0, s, t[0]], [0, 1, 0, t[1]], [-s, 0, c, t[2]]], np.float64)

    def cloud(m, n_bad=0, noise=0.002):
        prev = (rng.uniform(-4, 4, size=(m, 3)) * [1, 0.3, 3] + [0, 0, 12]).astype(np.float32)
        cur = (prev + [0.01, 0.0, -0.25] + rng.normal(scale=noise, size=(m, 3))).astype(np.float32)
        if n_bad:
            bad = rng.choice(m, n_bad, replace=False)
            cur[bad] += rng.normal(scale=2.0, size=(n_bad, 3)).astype(np.float32)
        return prev, cur

    good = T34([0.01, 0.0, -0.25], 0.01)
    nanT = good.copy(); nanT[1, 2] = np.nan
    third = np.pi / 3
    cases = [
        # name, odometer kwargs, skipped_frames, (prev, cur), transforms estimateAffine3D hands back in call order, Rodrigues vector
        ("defaults_ok", {}, 0, cloud(40), [good], [0, 0.01, 0]),
        ("defaults_7pts_min5", dict(min_matches=5), 0, cloud(7), [good], [0, 0.01, 0]),         # < 10 points: "rigidity" is set AND T returned
        ("defaults_7pts", {}, 0, cloud(7), [], [0, 0, 0]),                                      # < 10 points, < min_matches: None, "rigidity"
        ("rigid_ok", dict(rigidity_threshold=0.1), 0, cloud(60, 12), [good], [0, 0.01, 0]),
        ("rigid_too_few", dict(rigidity_threshold=0.02), 0, cloud(14, 9, noise=0.0005), [], [0, 0, 0]),
        ("rigid_9_min5", dict(rigidity_threshold=0.02, min_matches=5), 0, cloud(15, 6, noise=0.0005), [good], [0, 0.01, 0]),
        ("outlier_ok", dict(outlier_threshold=0.02), 0, cloud(50, 8), [good, T34([0.011, 0, -0.251], 0.0101)], [0, 0.0101, 0]),
        ("outlier_too_few", dict(outlier_threshold=0.0001), 0, cloud(12, 0, noise=0.05), [T34([3.0, 0, 0])], [0, 0, 0]),
        ("rigid_then_outlier", dict(rigidity_threshold=0.1, outlier_threshold=0.02), 0, cloud(80, 20), [good, good], [0, 0.01, 0]),
        ("nan", {}, 0, cloud(30), [nanT], [0, 0, 0]),
        ("bigdist_s0", {}, 0, cloud(30), [T34([0.9, 0.0, 1.2])], [0, 0.2, 0]),                   # |t| = 1.5 > 1
        ("bigdist_s2_passes", {}, 2, cloud(30), [T34([0.9, 0.0, 1.2])], [0, 0.2, 0]),            # 1.5 <= 3
        ("bigdist_s2_fails", {}, 2, cloud(30), [T34([3.0, 0.0, 0.5])], [0, 0.2, 0]),
        ("bigrot_s0", {}, 0, cloud(30), [T34([0.1, 0, 0.2], 1.2)], [0, 1.2, 0]),
        ("bigrot_s2_passes", {}, 2, cloud(30), [T34([0.1, 0, 0.2], 1.2)], [0, 1.2, 0]),
        ("bigrot_s2_fails", {}, 2, cloud(30), [T34([0.1, 0, 0.2], 3.15)], [0.1, 3.15, 0]),
        ("both_gates", {}, 0, cloud(30), [T34([2.0, 0, 0], 1.3)], [0, 1.3, 0]),                  # "bigrot" is written last
        ("dist_exactly_1", {}, 0, cloud(30), [T34([0.6, 0.0, 0.8])], [0, 0, 0]),                 # |t| == 1.0: not > 1
        ("rot_exactly_limit", {}, 0, cloud(30), [T34([0, 0, 0.1])], [third, 0, 0]),              # == pi/3: not >
        ("rot_just_over", {}, 0, cloud(30), [T34([0, 0, 0.1])], [np.nextafter(third, 4), 0, 0]),
        ("rigid_cause_then_bigdist", dict(min_matches=5), 1, cloud(8), [T34([2.5, 0, 0])], [0, 0, 0]),   # "rigidity" overwritten by "bigdist"
    ]
    out = {}
    for name, kw, skipped, (prev, cur), Ts, rvec in cases:
        od = new_odom(**kw)
        od.skipped_frames = skipped
        od.skip_cause = "init"
        queue, est_lens, rod_calls, masks = list(Ts), [], [], []

        def estimate(src, dst, force_rotation=False, _q=queue, _l=est_lens):
            assert force_rotation is True and len(src) == len(dst)
            _l.append(len(src))
            return _q.pop(0).copy(), 1.0

        def rodrigues(R, _r=rvec, _c=rod_calls):
            _c.append(np.array(R))
            return np.array(_r, np.float64).reshape(3, 1), None

        cv2.estimateAffine3D, cv2.Rodrigues = estimate, rodrigues
        real_filter = od.rigid_body_filter

        def rigid(a, b, _m=masks):
            m = real_filter(a, b)
            _m.append(np.asarray(m))
            return m

        od.rigid_body_filter = rigid
        with np.errstate(all="ignore"):
            ret = od.point_cloud_transform(prev.copy(), cur.copy())
        assert not queue, name                       # every scripted transform was asked for
        del cv2.estimateAffine3D, cv2.Rodrigues
        pre = name + "__"
        out[pre + "prev"], out[pre + "cur"] = prev, cur
        out[pre + "kw"] = np.array([kw.get("rigidity_threshold", 0), kw.get("outlier_threshold", 0), kw.get("min_matches", 10), skipped], np.float64)
        out[pre + "Ts"] = np.array(Ts, np.float64).reshape(-1, 3, 4)
        out[pre + "rvec"] = np.array(rvec, np.float64)
        out[pre + "ret_none"] = np.array(ret is None)
        if ret is not None:
            out[pre + "ret"] = np.asarray(ret, np.float64)
        out[pre + "skip_cause"] = np.array(od.skip_cause)
        out[pre + "est_lens"] = np.array(est_lens, np.int64)       # points handed to each estimateAffine3D call
        out[pre + "n_rodrigues"] = np.array(len(rod_calls))
        out[pre + "mask"] = masks[0] if masks else np.zeros(0, np.int64)
    out["names"] = np.array([c[0] for c in cases])
    np.savez(os.path.join(HERE, "g8_point_cloud_transform.npz"), **out)


if __name__ == "__main__":
    g1_feature_mask(); g2_bilinear(); g3_rigid(); g4_outlier(); g5_state_machine(); g6_rot2rpy()
    g7_point_clouds(); g8_point_cloud_transform()
    print("golden fixtures written to", HERE)
