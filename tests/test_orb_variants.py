"""The two details of orb.cpp the oracle recalls rather than reads, QUANTIFIED: so that whoever runs
tests/test_cv2_crosscheck.py against a real OpenCV first knows what a mismatch there would mean.  CPU only (oracle twice)."""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, ".."))
sys.path.insert(0, HERE)
from oracle import oracle as O          # noqa: E402
import orb_variants as V                # noqa: E402

BASELINE_DIMS = (640, 480, 1280, 720, 2048, 1536, 1920, 1080, 3840, 2160)


def test_level_size_rules_differ_only_on_the_committed_widths_and_on_no_baseline_size():
    """cvRound(cols * (1.f / scale)) against cvRound(cols / scale), every width 1 .. 8192 at every level: 281 of 65 536
    combinations differ (273 of them at level 1: widths = 9 mod 12, where cols / 1.2f lands on x.5 exactly in float and the
    product with the rounded reciprocal does not) -- and none of them is a dimension of a BASELINE config."""
    rows = [list(r) for r in V.level_size_differences()]
    want = json.load(open(os.path.join(HERE, "golden", "orb_level_size_rule_differences.json")))["rows"]
    assert rows == want
    assert len(rows) == 281
    assert not [r for r in rows if r[1] in BASELINE_DIMS]
    # the two rules never differ by more than one pixel
    assert all(abs(r[2] - r[3]) == 1 for r in rows)


def test_cosf_against_double_cos_changes_almost_nothing():
    """The descriptor's rotation: (float)cos((double)a) [built] against cosf(a) [what cos(float) is under libstdc++].  Detection
    does not see it; of the descriptor bits of 6 frames (300 features each) at most a handful may move.  At BASELINE size
    (profiles/r05_orb_variants.json, tests/orb_variants.py): config 2, 5 frames x 500: 0 bits, matches and chained pose
    identical; config 5, 5 frames x 8000 = 10.2 M bits: ONE bit, no match changed."""
    from openvo_amd.synth import Corridor
    c = Corridor("T0")
    bits = total = 0
    for k in range(6):
        img, _ = c.pair(k)
        a = O.orb_detect_and_compute(img, None, 300)
        with O.orb_variant(2):
            b = O.orb_detect_and_compute(img, None, 300)
        assert O.lib().vo_ref_orb_get_variant() == 0
        assert np.array_equal(a["xy"], b["xy"]) and np.array_equal(a["angle"], b["angle"])
        assert np.array_equal(a["response"], b["response"]) and np.array_equal(a["octave"], b["octave"])
        bits += int(np.unpackbits(a["desc"] ^ b["desc"]).sum())
        total += a["desc"].size * 8
    assert total > 100000 and bits <= total * 1e-5


def test_the_committed_measurement_says_what_design_quotes():
    m = json.load(open(os.path.join(HERE, "..", "profiles", "r05_orb_variants.json")))
    assert m["level_size"]["differing"] == 281 and m["level_size"]["baseline_dims_affected"] == []
    assert m["cosf_c2"]["bits"] == 0 and m["cosf_c2"]["matches_changed"] == 0 and m["cosf_c2"]["pose_max_abs_diff"] == 0.0
    assert m["cosf_c5"]["bits"] <= 2 and m["cosf_c5"]["matches_changed"] == 0
