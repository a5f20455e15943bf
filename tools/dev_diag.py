"""Development check of the diagonal-sweep schedule on the GPU (not a test: tests/test_gpu_configs.py has those).

Compares VO_DIAG=1 with the line schedule bit for bit on a list of sizes / modes, prints where they differ, and times
the aggregation stages of one pair alone (HIP events around the stage, 10 repetitions).
Usage: python tools/dev_diag.py [quick|full]
"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from openvo_amd import _native  # noqa: E402
from openvo_amd.synth import Corridor  # noqa: E402


def run(c, L, R, p, mode, env, reps=0):
    for k in ("VO_DIAG", "VO_DIAG_WAVES", "VO_WE_FUSE", "VO_DIAG_DEBUG"):
        os.environ.pop(k, None)
    os.environ.update(env)
    h, w = L.shape
    ctx = _native.Context(0, max(w, 64), max(h, 64), max(16, (p["numDisparities"] + 15) // 16 * 16), 64)
    ctx.set_sgbm(p, mode)
    out = ctx.sgbm_compute_host(L, R)
    st = ctx.sgbm_sweep_status()
    tm = None
    if reps:
        ctx.enable_timing(True, ["sgbm_cost", "sgbm_agg", "sgbm_wta", "sgbm_post"])
        ctx.timings(reset=True)
        t0 = time.perf_counter()
        for _ in range(reps):
            out2 = ctx.sgbm_compute_host(L, R)
        wall = (time.perf_counter() - t0) / reps * 1e3
        assert np.array_equal(out, out2) or env.get("VO_DIAG_DEBUG"), "not deterministic"
        t = ctx.timings(reset=True)
        tm = {k: round(v[0] / max(v[1], 1), 4) for k, v in t.items() if v[1]}
        tm["wall_ms"] = round(wall, 3)
    if reps and env.get("VO_DIAG") == "1":
        tm["timeline"] = timeline(ctx.sgbm_sweep_stats(1 if mode == 1 else 0), w - p["numDisparities"], h, 4 * int(env.get("VO_DIAG_WAVES", "15")) if p["numDisparities"] <= 128 else 28)
    if reps and env.get("VO_DIAG") == "1" and os.environ.get("DG_TRACE"):
        tr = ctx.sgbm_sweep_stats(1).astype(np.int64) & 0xFFFFFFFF
        t = (tr[:900] & 0xFFFFFF).reshape(3, 300)
        sp = (tr[:900] >> 24).reshape(3, 300)
        base = t[t > 0].min() if (t > 0).any() else 0
        for q in range(3):
            print("trace strip +%d rows 100..139: t(us) %s" % (q, [round((int(x) - int(base)) / 100.0, 1) for x in t[q, :40]]))
            print("   spins %s" % [int(x) for x in sp[q, :40]])
        print("   row period strip+0 (us): mean %.3f  strip+1 minus strip+0 at the same row: mean %.3f us" % (
            float(np.diff(t[0, :290]).mean()) / 100.0, float((t[1, :290] - t[0, :290]).mean()) / 100.0))
    ctx.close()
    return out, st, tm


def timeline(words, W1, H, UW):
    """per-strip {start, end, failed polls, ticks waiting} (100 MHz ticks) -> a summary in microseconds"""
    ns = -(-(W1 + H - 1) // UW)
    st = words[8:8 + 8 * ns].reshape(ns, 8).astype(np.int64) & 0xFFFFFFFF
    t0 = st[:, 0].min()
    rows = np.array([min(H, W1 + H - 1 - j * UW) - max(0, H - 1 - (j * UW + UW - 1)) for j in range(ns)])
    beg, end, spins, wait = (st[:, 0] - t0) / 100.0, (st[:, 1] - t0) / 100.0, st[:, 2], st[:, 3] / 100.0
    busy = end - beg - wait
    sel = rows >= 100
    return {"strips": ns, "span_us": round(float(end.max()), 1), "row_us_busy": round(float((busy[sel] / rows[sel]).mean()), 3),
            "wait_us_total": round(float(wait.sum()), 1), "wait_us_max": round(float(wait.max()), 1), "failed_polls": int(spins.sum()), "misses": int(st[:, 4].sum()), "misses_by_strip": [int(x) for x in st[::max(1, ns // 16), 4]],
            "end_us_by_strip": [int(e) for e in end[::max(1, ns // 16)]], "wait_us_by_strip": [int(x) for x in wait[::max(1, ns // 16)]],
            "begin_us_by_strip": [int(e) for e in beg[::max(1, ns // 16)]],
            "tail": [(int(j), int(rows[j]), round(float(end[j] - beg[j]), 1), round(float(wait[j]), 1), int(st[j, 4])) for j in range(max(0, ns - 8), ns)]}


def main():
    full = len(sys.argv) > 1 and sys.argv[1] == "full"
    cases = [("T0", 0, 48, None), ("C1", 0, 64, None), ("C1", 1, 64, None), ("C1", 0, 112, None), ("C1", 1, 112, (632, 471)),
             ("C1", 0, 64, (200, 59)), ("C1", 0, 96, None), ("C1", 0, 32, None)]
    if full:
        cases += [("C2", 0, 128, None), ("C4", 1, 256, None)]
    bad = 0
    for name, mode, nd, crop in cases:
        c = Corridor(name)
        L, R = c.pair(4)
        if crop:
            L, R = np.ascontiguousarray(L[:crop[1], :crop[0]]), np.ascontiguousarray(R[:crop[1], :crop[0]])
        p = c.sgbm_params(mode)
        p["numDisparities"] = nd
        ref, _, _ = run(c, L, R, p, mode, {"VO_DIAG": "0", "VO_WE_FUSE": "0"})
        for waves in ("15", "7"):
            got, st, _ = run(c, L, R, p, mode, {"VO_DIAG": "1", "VO_DIAG_WAVES": waves})
            ne = got != ref
            print("%s mode %d D %d %s waves %s: %s  status %d  differing %d of %d" % (
                name, mode, nd, L.shape, waves, "OK" if not ne.any() and st == 0 else "MISMATCH", st, int(ne.sum()), ne.size), flush=True)
            if ne.any():
                bad += 1
                ys, xs = np.nonzero(ne)
                print("   rows %d..%d cols %d..%d; first: %s" % (ys.min(), ys.max(), xs.min(), xs.max(),
                      [(int(y), int(x), int(got[y, x]), int(ref[y, x])) for y, x in list(zip(ys, xs))[:8]]))
                print("   per-row counts (first 12 rows with any):", [(int(y), int(n)) for y, n in zip(*np.unique(ys, return_counts=True))][:12])
    for name, mode in (("C2", 0),) + ((("C4", 1),) if full else ()):
        c = Corridor(name)
        L, R = c.pair(4)
        p = c.sgbm_params(mode)
        for env in ({"VO_DIAG": "0", "VO_WE_FUSE": "0"}, {"VO_DIAG": "0", "VO_WE_FUSE": "1"}, {"VO_DIAG": "1", "VO_DIAG_WAVES": "15"},
                    {"VO_DIAG": "1", "VO_DIAG_WAVES": "7"}, {"VO_DIAG": "1", "VO_DIAG_WAVES": "7", "VO_DIAG_DEBUG": "1"},
                    {"VO_DIAG": "1", "VO_DIAG_WAVES": "7", "VO_DIAG_DEBUG": "3"}, {"VO_DIAG": "1", "VO_DIAG_WAVES": "15", "VO_DIAG_DEBUG": "3"}):
            if name == "C4" and env.get("VO_DIAG_WAVES") == "15":
                continue
            _, st, tm = run(c, L, R, p, mode, env, reps=10)
            print(name, env, "status", st, tm, flush=True)
    print("mismatching cases:", bad)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
