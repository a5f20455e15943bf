#!/usr/bin/env python3
"""What the two recalled-not-read details of orb.cpp weigh.  CPU only: the oracle twice.

(a) pyramid level size: cvRound(cols * (1.f / scale)) [built] against cvRound(cols / scale): every width 1 .. 8192 x 8 levels.
(b) descriptor rotation: (float)cos((double)a) [built] against cosf(a): descriptor bits, kNN-2 + ratio matches and the chained
    pose on frames of config 2 (stereo 1280 x 720, 500 features) and config 5 (mono 1920 x 1080, 8000 features).

    python tests/orb_variants.py [--pairs 6] [--json out.json]
"""
import argparse
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from oracle import oracle as O                                     # noqa: E402
from oracle.odometer import RefStereoCamera, RefStereoOdometer     # noqa: E402
from openvo_amd.synth import Corridor                              # noqa: E402
from openvo_amd import calib                                       # noqa: E402


def _rig(c):
    """Q and the left ROI of the corridor's ideal rig, from the host-side calibration code (no GPU)."""
    rp = c.rect_params()
    r = calib.stereo_rectify(c.K(), c.dist(), c.K(), c.dist(), (c.w, c.h), rp["R"], rp["T"])
    return r[4], tuple(int(v) for v in r[5])


def level_size_differences(max_w=8192):
    out = []
    for lvl in range(8):
        for w in range(1, max_w + 1):
            a = O.orb_level_size(w, w, lvl)[0]
            with O.orb_variant(1):
                b = O.orb_level_size(w, w, lvl)[0]
            if a != b:
                out.append((lvl, w, a, b))
    return out


def _matches(d1, d2, ratio=0.8):
    idx, dist = O.bf_knn2_hamming(d1, d2)
    q, t = O.ratio_filter(idx, dist, ratio)
    return set(zip(q.tolist(), t.tolist()))


def desc_diff(a, b):
    """keypoint sets must be identical (the rotation does not enter detection); returns differing bits / descriptors"""
    assert np.array_equal(a["xy"], b["xy"]) and np.array_equal(a["angle"], b["angle"])
    x = np.bitwise_xor(a["desc"], b["desc"])
    bits = int(np.unpackbits(x).sum())
    return bits, int((x.any(axis=1)).sum()), len(a["desc"])


def c5(pairs):
    c = Corridor("C5")
    res = dict(bits=0, descriptors=0, total=0, matches_changed=0, matches_total=0)
    prev = None
    for k in range(pairs + 1):
        img, _ = c.pair(k)
        a = O.orb_detect_and_compute(img, None, 8000)
        with O.orb_variant(2):
            b = O.orb_detect_and_compute(img, None, 8000)
        bits, nd, n = desc_diff(a, b)
        res["bits"] += bits; res["descriptors"] += nd; res["total"] += n
        if prev is not None:
            ma, mb = _matches(prev[0]["desc"], a["desc"]), _matches(prev[1]["desc"], b["desc"])
            res["matches_changed"] += len(ma ^ mb); res["matches_total"] += len(ma)
        prev = (a, b)
    return res


def c2(pairs):
    c = Corridor("C2")
    runs = []
    for flags in (0, 2):
        with O.orb_variant(flags):
            cam = RefStereoCamera(*_rig(c), c.sgbm_params())
            odo = RefStereoOdometer(cam, nfeatures=500, rigidity_threshold=0.1, outlier_threshold=0.02,
                                    preprocessed_frames=True)
            frames = []
            for k in range(pairs + 1):
                L, R = c.pair(k)
                ok = odo.update(L, R)
                frames.append(dict(ok=ok, desc=odo.cur["desc"].copy(), xy=odo.cur["kps"]["xy"].copy(),
                                   angle=odo.cur["kps"]["angle"].copy(), pose=odo.c_T_w.copy()))
            runs.append(frames)
    res = dict(bits=0, descriptors=0, total=0, matches_changed=0, matches_total=0, pose_max_abs_diff=0.0, decisions_equal=True)
    for k, (fa, fb) in enumerate(zip(*runs)):
        bits, nd, n = desc_diff(fa, fb)
        res["bits"] += bits; res["descriptors"] += nd; res["total"] += n
        res["decisions_equal"] &= fa["ok"] == fb["ok"]
        res["pose_max_abs_diff"] = max(res["pose_max_abs_diff"], float(np.abs(fa["pose"] - fb["pose"]).max()))
        if k:
            ma = _matches(runs[0][k - 1]["desc"], fa["desc"]); mb = _matches(runs[1][k - 1]["desc"], fb["desc"])
            res["matches_changed"] += len(ma ^ mb); res["matches_total"] += len(ma)
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--pairs", type=int, default=6)
    ap.add_argument("--json")
    a = ap.parse_args()
    d = level_size_differences()
    out = dict(level_size=dict(widths_tested=8192, levels=8, differing=len(d),
                               baseline_dims_affected=[x for x in d if x[1] in (640, 480, 1280, 720, 2048, 1536, 1920, 1080)]),
               cosf_c2=c2(a.pairs), cosf_c5=c5(a.pairs), pairs=a.pairs)
    print(json.dumps(out, indent=1))
    if a.json:
        json.dump(out, open(a.json, "w"), indent=1)


if __name__ == "__main__":
    main()
