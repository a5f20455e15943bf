"""Which SGBM stage limits the throughput of the look-ahead engines?  Development aid.

Runs disparity only (no ORB, no pose) for N staged C2 pairs through the engines, once complete and once with each stage
skipped (VO_DIAG_DEBUG bits: results are then garbage, only the clock counts); the time a stage adds under concurrency is
the difference.  One process per variant (the knob is read at context creation).
Usage: python tools/stage_ablation.py [engine counts, e.g. 1,2,4,8,12]"""
import os, subprocess, sys, time

if len(sys.argv) > 1 and sys.argv[1] == "child":
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from openvo_amd import _native
    from openvo_amd.synth import Corridor
    c = Corridor("C2")
    N = 240
    ctx = _native.Context(0, c.w, c.h, c.D, 500)
    ctx.set_sgbm(c.sgbm_params(0), 0)
    ctx.stage_pairs([c.pair(i) for i in range(24)])
    for i in range(24):
        ctx.prefetch_staged_pair(i % 24, i % 24, True)
    ctx.synchronize()
    t0 = time.perf_counter()
    for i in range(N):
        ctx.prefetch_staged_pair(i % 24, i % 24, True)
    ctx.synchronize()
    dt = time.perf_counter() - t0
    print("%-28s %.3f ms/pair  (%.0f pairs/s)" % (os.environ.get("TAG", ""), 1e3 * dt / N, N / dt))
    sys.exit(0)

VARIANTS = (("complete", 0), ("no cost", 4), ("no W+E", 8), ("no diagonal", 16), ("no post", 32), ("only cost", 56), ("only W+E", 52),
            ("only diagonal", 44), ("only post", 28))
engines = sys.argv[1].split(",") if len(sys.argv) > 1 else ["12"]
for tag, dbg in VARIANTS:
    for n in engines:
        # the stage switches exist in the test-only build alone (libvo355_hooks.so)
        hooks = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "openvo_amd", "libvo355_hooks.so")
        e = dict(os.environ, TAG="%s, %s engines" % (tag, n), VO_DIAG_DEBUG=str(dbg), VO_ENGINES=n, VO355_LIB=os.path.abspath(hooks))
        subprocess.run([sys.executable, os.path.abspath(__file__), "child"], env=e, timeout=300)
