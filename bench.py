#!/usr/bin/env python3
"""bench.py -- stereo frame-pairs/sec of the openVO hot path on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W
    (N > 1: one rank per GPU, started by any launcher that sets RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR /
     MASTER_PORT, e.g. `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...`;
     nothing of torch is imported here -- the ranks meet on a local socket and gather poses over RCCL)

One step = one StereoOdometer.update() on one 1280x720 stereo pair of the synthetic corridor
sequence (BASELINE config 2, "C2": SGBM D=128 5-path + ORB 500 + Hamming kNN/ratio + 3-D lookup +
rigid-clique filter + Umeyama), inputs already resident in HBM.  Each rank owns one GPU and a
contiguous chunk of the sequence (weak scaling); the only exchange is the final gather of the relative
poses (RCCL all-gather bound behind the C ABI, include/vo355.h vo_mgpu_*).  Rank 0 prints ONE JSON line
with the contract fields plus `roofline` (dominant kernel: the SGBM path aggregation, timed with HIP
events on the library's own stream) and `cpu_baseline` (the CPU path timed on this box's host cores,
N = 1 only: one thread and all cores; a real cv2 when one is importable, the oracle port otherwise).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

# (GPU_MAX_HW_QUEUES is set by openvo_amd at import, before any HIP runtime starts: 24, or this process' share with VO_SHARE_GPU)

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# The reference's defaults are rigidity_threshold = outlier_threshold = 0 (stereo_odometer.py:14-15); the bench
# switches the reference's own optional stages ON (more work per pair, not less): see `deviations` in the output.
ODO_KW = dict(nfeatures=500, match_threshold=0.8, rigidity_threshold=0.1, outlier_threshold=0.02,
              preprocessed_frames=True, min_matches=10)
SHADER_GHZ = 2.4            # measured under load: tools/clock_probe.hip (2.38-2.41)
HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
DEVIATIONS = {
    "odometer": "rigidity_threshold=0.1, outlier_threshold=0.02 instead of the reference defaults 0 / 0: with the plain fit a "
                "handful of gross mismatches drives the pose metres off (OpenCV alike); the optional stages ADD the clique "
                "filter and the outlier pass to every measured pair",
    "scene": "back wall world-fixed at z = 400 m instead of SURVEY 8(d)'s wall moving with the camera (features on a wall "
             "that moves with the camera contradict the camera motion); everything else as specified",
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=240)
    ap.add_argument("--warmup", type=int, default=12)
    ap.add_argument("--workload", default="C2")
    ap.add_argument("--no-events", action="store_true", help="diagnostic: do not record HIP events in the timed region")
    ap.add_argument("--from-host", action="store_true",
                    help="diagnostic: hand host numpy images to the odometer every step (StereoOdometer.run: pinned "
                         "staging + async upload ahead) instead of HBM-resident inputs -- the PCIe-inclusive rate")
    ap.add_argument("--ndisp", type=int, default=0, help="diagnostic: override numDisparities (changes the workload!)")
    ap.add_argument("--cpu-pairs", type=int, default=8, help="pairs the single-thread CPU baseline is timed on (0 = skip both CPU legs)")
    ap.add_argument("--no-post", action="store_true", help="skip the untimed post-passes (per-stage breakdown, from-host rate, other configs)")
    ap.add_argument("--no-other", action="store_true", help="skip the C4 / C5 runs attached to the C2 line as `other_configs`")
    ap.add_argument("--repeats", type=int, default=5,
                    help="cold-start windows of --steps pairs each, back to back on consecutive frames; `value` is their median "
                         "(capped so that at most 600 frames are rendered)")
    ap.add_argument("--mix-stages", action="store_true",
                    help="diagnostic: one more steady pass with HIP events around EVERY stage on its own stream -> `stage_ms_per_pair_in_mix` "
                         "(a stage's latency beside the other pairs' kernels; the events themselves cost some throughput)")
    ap.add_argument("--steady", type=int, default=960,
                    help="pairs of the untimed-for-`value` steady-state pass reported as `steady_state` (0 = skip)")
    args = ap.parse_args()

    from openvo_amd import sharding

    group, device = sharding.init_from_env()
    if args.workload == "C5":
        out = bench_c5(args, group, device, args.steps, args.warmup)
    else:
        out = bench_stereo(args, group, device, args.workload, args.steps, args.warmup, light=False)
    if group.rank == 0:
        if args.workload == "C2" and group.world == 1 and not (args.no_post or args.no_other or args.from_host or args.ndisp):
            # BASELINE configs 4 and 5 in the same process, after the headline's work is done and its context is closed:
            # their own complete JSON objects (shorter windows), so that the driver's one command observes them too
            other = {}
            for name, fn in (("C4", lambda: bench_stereo(args, group, device, "C4", 12, 4, light=True)),
                             ("C5", lambda: bench_c5(args, group, device, 200, 12))):
                try:
                    other[name] = fn()
                except Exception as e:                      # a failure here must not cost the headline line
                    other[name] = {"error": "%s: %s" % (type(e).__name__, e)}
            out["other_configs"] = other
        print(json.dumps(out))
    group.barrier()
    group.close()


def bench_stereo(args, group, device, workload, K, W, light):
    """K timed StereoOdometer.update() steps of one stereo workload (C2, C4, C1 ...) on this rank's GPU; returns the JSON
    object on rank 0 (None elsewhere).  light = no post-passes, no CPU leg (the attached other_configs)."""
    from openvo_amd import StereoCamera, StereoOdometer, sharding
    from openvo_amd.synth import Corridor
    rank, world = group.rank, group.world
    # (a rehearsal with several ranks on one GPU: VO_SHARE_GPU=<ranks per GPU> in the environment, read by openvo_amd at import)
    if args.gpus != world and rank == 0 and world > 1:
        print("warning: --gpus %d but WORLD_SIZE %d" % (args.gpus, world), file=sys.stderr)

    c = Corridor(workload)
    # config 4 is the 8-path (MODE_HH) cost-volume stress case; every other workload runs the reference's 5-path default
    sgbm = c.sgbm_params(mode=1) if workload == "C4" else c.sgbm_params()
    if args.ndisp and not light:
        sgbm["numDisparities"] = args.ndisp
    cam = StereoCamera(c.K(), c.dist(), c.K(), c.dist(), c.rect_params(), sgbm, (c.w, c.h), device=device,
                       max_keypoints=ODO_KW["nfeatures"])
    odo = StereoOdometer(cam, **ODO_KW)
    # this rank's frames: W warm-up frames (they also provide the halo), then R windows of K timed frames each on consecutive
    # frames (`value` = the median window).  The steady-state pass walks the same frames forwards and backwards (consecutive
    # frames either way: every step is a real pair of the sequence with the same work) so that nothing more has to be rendered.
    R = 1 if (light or args.from_host) else max(1, min(args.repeats, max(1, (600 - W) // K)))
    n_unique = W + R * K
    first = rank * R * K
    frames = c.pairs(first, n_unique)
    S_steady = 0 if (light or args.no_post or args.from_host or args.ndisp) else max(0, args.steady)
    walk = list(range(n_unique))
    while S_steady and len(walk) < W + S_steady:          # 0 .. n-1, n-2 .. 0, 1 .. n-1, ...
        walk += list(range(n_unique - 2, -1, -1)) + list(range(1, n_unique))
    walk = walk[:max(n_unique, W + S_steady)]
    staged = cam.stage_pairs([frames[i] for i in walk])   # inputs resident in HBM before any clock starts
    ctx = cam._ctx

    def sync_all():
        ctx.synchronize()                     # every stream of this rank's context (hipStreamSynchronize)
        group.barrier()

    # the interpreter's cyclic GC must not fire inside the timed region: collect now, then keep it off
    import gc
    gc.collect()
    gc.freeze()
    gc.disable()
    if args.from_host and not light:
        for ok in odo.run(frames[:W]):          # warm the path that is timed (pinned staging is allocated on first use)
            pass
    else:
        for i in range(W):
            odo.update(staged[i], None)
    # HIP events (recorded on the library's stream, resolved after the run) around the dominant
    # kernel only: event packets around every small stage would perturb the throughput measured
    ctx.enable_timing(not args.no_events, stages=["sgbm_wta"])
    ctx.timings(reset=True)
    rel, acc, dts, clocks = [], [], [], []

    def window(od, lo, n, record):
        """n update() steps on staged[lo : lo + n] from a cold start: every stream drained and every look-ahead result
        dropped before the clock starts, every stream drained before it stops; returns this rank's seconds."""
        od.reset_lookahead()       # nothing computed before the clock starts may be used inside the timed region
        cam.lookahead_stop = lo + n    # ... and nothing beyond the window is started inside it: exactly n pairs of work
        sync_all()
        t0 = time.perf_counter()
        if args.from_host and not light:
            before = od.c_T_w
            for ok in od.run(frames[lo:lo + n]):
                acc.append(bool(ok))
                rel.append(sharding.relative_from_chain(before, od.c_T_w) if ok else np.eye(4))
                before = od.c_T_w
        else:
            for i in range(lo, lo + n):
                before = od.c_T_w
                ok = od.update(staged[i], None)
                if record:
                    acc.append(bool(ok))
                    rel.append(sharding.relative_from_chain(before, od.c_T_w) if ok else np.eye(4))
        ctx.synchronize()                         # every stream of this rank's context has drained: this rank's n steps are done
        dt = time.perf_counter() - t0             # (the MAX over ranks is taken below; the closing barrier is not part of any rank's work)
        if record:
            clocks.append(round(ctx.shader_clock(200), 0))   # after the clock has stopped: the shader clock this window ended at
        group.barrier()
        cam.lookahead_stop = None
        return dt

    for r in range(R):                            # R windows of exactly K steps each, on consecutive frames
        dts.append(window(odo, W + r * K, K, True))
    tm = ctx.timings(reset=True)
    ctx.enable_timing(False)
    odo.reset_lookahead()
    sweep_err = ctx.sgbm_sweep_status()
    # the same K-step window with the reference's own default odometer (rigidity_threshold = outlier_threshold = 0,
    # /root/reference/src/openVO/stereo_odometer.py:14-15) and a long steady-state pass -- neither is `value`
    dt_default = dt_steady = None
    steady_acc = 0
    dodo = sodo = probe = None
    if not (light or args.no_post or args.from_host):
        kw = dict(ODO_KW, rigidity_threshold=0, outlier_threshold=0)
        dodo = StereoOdometer(cam, **kw)
        for i in range(W):
            dodo.update(staged[i], None)
        dt_default = window(dodo, W, K, False)
        dodo.reset_lookahead()
    if S_steady:
        sodo = StereoOdometer(cam, **ODO_KW)
        for i in range(W):
            sodo.update(staged[i], None)
        sodo.reset_lookahead()
        cam.lookahead_stop = W + S_steady
        sync_all()
        t0 = time.perf_counter()
        for i in range(W, W + S_steady):
            steady_acc += bool(sodo.update(staged[i], None))
        ctx.synchronize()
        dt_steady = time.perf_counter() - t0
        group.barrier()
        cam.lookahead_stop = None
        sodo.reset_lookahead()
        sweep_err |= ctx.sgbm_sweep_status()
    tmix, mix_rate = None, None
    if S_steady and args.mix_stages:
        modo = StereoOdometer(cam, **ODO_KW)
        for i in range(W):
            modo.update(staged[i], None)
        modo.reset_lookahead()
        ctx.enable_timing(True)
        ctx.timings(reset=True)
        sync_all()
        t0 = time.perf_counter()
        for i in range(W, W + S_steady):
            modo.update(staged[i], None)
        ctx.synchronize()
        mix_rate = S_steady / (time.perf_counter() - t0)
        tmix = ctx.timings(reset=True)
        ctx.enable_timing(False)
        modo.reset_lookahead()
        modo = None
    gc.enable()
    schedule = {1: "diag", 2: "diag_ragged", 3: "unfused"}.get(ctx.sgbm_last_schedule(), "?")

    tb, tb_eng, nb, from_host_rate, copy_gbs = None, None, 0, None, None
    if not (args.no_post or light):
        # the box's own streaming-copy ceiling: one cost volume's worth of bytes copied between two of the
        # context's volumes, plain and non-temporal, HIP events around 20 repetitions -- untimed, after the measured region
        ctx.synchronize()
        copy_gbs = {"plain": round(ctx.measure_copy(0, 20, False), 1), "nontemporal": round(ctx.measure_copy(0, 20, True), 1)}
    if not args.no_post:
        # per-stage breakdown: a short untimed post-pass with every stage timed and the look-ahead engines off, i.e. one
        # pair at a time with each kernel alone on the GPU -- the same condition a rocprofv3 kernel trace imposes (it
        # serialises dispatches).  The roofline block's top-level figures come from here.
        ctx.enable_timing(True)
        nb = min(8, K)
        la = cam.lookahead
        cam.reset_lookahead()
        cam.lookahead = 0
        probe = StereoOdometer(cam, **ODO_KW)
        for i in range(W + K - nb - 1, W + K):
            probe.update(staged[i], None)
        tb = ctx.timings(reset=True)
        # the same again with ONE pair ahead on a look-ahead engine: the kernels of the streamed path (the diagonal sweep runs
        # its wide-strip variant there, the narrow-strip one in a synchronous call: sgbm.hip, launch_diag), still one pair's
        # launches at a time -- beside them only the previous pair's four short pose kernels
        cam.reset_lookahead()
        cam.lookahead = 1
        probe = StereoOdometer(cam, **ODO_KW)
        for i in range(W + K - nb - 1, W + K):
            probe.update(staged[i], None)
        ctx.synchronize()
        tb_eng = ctx.timings(reset=True)
        ctx.enable_timing(False)
        cam.lookahead = la
    if not (args.no_post or light) and not args.from_host and world == 1:
        # PCIe-inclusive rate (never the reported value): the frames that follow in the same sequence, handed over as host
        # numpy arrays -- 8 untimed pairs, then 480 timed ones whatever K is (a 20-pair window would mostly time the start
        # of the staging thread and the pipeline's fill): 96 further frames of the sequence, walked forwards and backwards
        nh, nr = 480, 96
        hframes = c.pairs(first + n_unique, 8 + nr)
        hwalk = list(range(nr))
        while len(hwalk) < nh:
            hwalk += list(range(nr - 2, -1, -1)) + list(range(1, nr))
        hwalk = hwalk[:nh]
        # (the odometers of the earlier passes still own two frame slots each: give them back, the host path's look-ahead wants them)
        odo = dodo = sodo = probe = None
        hodo = StereoOdometer(cam, **ODO_KW)
        cam.reset_lookahead()
        for ok in hodo.run(hframes[:8]):
            pass
        ctx.synchronize()
        th = time.perf_counter()
        for ok in hodo.run(hframes[8 + i] for i in hwalk):
            pass
        ctx.synchronize()
        from_host_rate = nh / (time.perf_counter() - th)

    dt_windows = [group.all_reduce_max(d) for d in dts]      # per window: max over ranks of the timed region
    dt_max = float(np.median(dt_windows))                    # `value` comes from the median window
    if dt_default is not None:
        dt_default = group.all_reduce_max(dt_default)
    if dt_steady is not None:
        dt_steady = group.all_reduce_max(dt_steady)
    # final pose gather (the path's only exchange): 16 float64 + accept flag per frame
    all_rel, all_ok = group.gather_relative(np.array(rel), np.array(acc, np.float64))

    out = None
    if rank == 0:
        total_pairs = K * world
        value = total_pairs / dt_max
        cells, npaths = ctx.sgbm_last_geometry()
        P = 8 if workload == "C4" else 5
        alg_bytes = 2.0 * cells * npaths          # the int16 cost volume read once per direction the kernel covers (3)
        survey_bytes = 2.0 * cells * (1 + P)      # A_sgbm = 2 B * V * (1 + P) for the whole pair
        traffic, per_pair, traffic_rev, valu_pp, traffic_stale = None, None, "unknown", None, None
        tf = os.path.join(ROOT, "profiles", "traffic_%s.json" % workload)
        if os.path.exists(tf):
            try:
                tj = json.load(open(tf))
                traffic, per_pair = tj.get("dominant_kernel_bytes_per_launch"), tj.get("sgbm_bytes_per_pair")
                valu_pp = tj.get("valu_wave_instructions_per_pair")
                traffic_rev = tj.get("source", "unknown")
                from openvo_amd._native import csrc_digest
                traffic_stale = tj.get("csrc_digest") != csrc_digest()     # the counters were taken from other kernel sources (or carry no stamp)
                if not isinstance(per_pair, (int, float)):        # (a profile file of an older layout)
                    per_pair = None
            except Exception:
                traffic = per_pair = None
        roof = {"bound": "hbm",
                "kernel": "k_sgbm_diag (forward diagonal sweep: NW / N / NE of the %d directions + the W+E volume + winner-take-all; "
                          "schedule '%s' on every one of the %d timed pairs)" % (P, schedule, R * K),
                "peak": HBM_PEAK_GBS, "unit": "GB/s", "traffic": traffic, "bytes_per_launch": alg_bytes,
                "traffic_source": ("committed file profiles/traffic_%s.json, NOT measured in this run (PMC counters need rocprofv3 "
                                   "around the process): %s" % (workload, traffic_rev)) if traffic is not None else None,
                "traffic_stale": traffic_stale,      # True: profiles/traffic_<workload>.json was taken from other kernel sources than these
                "schedule_counts": {schedule: R * K}}
        if tb is not None:
            def alone(t):
                ms, n = t["sgbm_wta"]
                sec = (ms / 1e3) / max(n, 1)
                return sec, int(n), (alg_bytes / sec / 1e9 if sec > 0 else 0.0)
            # top level = the variant the timed region runs (streamed through a look-ahead engine), one pair's launches at a time:
            # what `rocprofv3 --kernel-trace --stats` of this command reports for the kernel (profiles/r04_kernel_stats_*.csv)
            e_s, e_n, e_ach = alone(tb_eng if tb_eng is not None and tb_eng["sgbm_wta"][1] else tb)
            s_s, s_n, s_ach = alone(tb)
            roof.update({"achieved": round(e_ach, 2), "frac": round(e_ach / HBM_PEAK_GBS, 5), "launch_us": round(e_s * 1e6, 2), "launches": e_n,
                         "condition": "the launch with one pair at a time on a look-ahead engine (HIP events on its stream; the post kernel's "
                                      "~10 us included).  Up to 128 disparities the streamed path runs the sweep in WIDER strips (11 compute "
                                      "waves, one 12-wave workgroup per CU on ~28 CUs): slower as a launch than the narrow strips of a "
                                      "synchronous call (`sync_call`), faster as a job (+8 %: the other pairs' kernels no longer share SIMDs "
                                      "with sweep waves; DESIGN 4b).  The kernel is a latency chain by design; the job's rate is in `aggregate`"})
            roof["sync_call"] = {"launch_us": round(s_s * 1e6, 2), "launches": s_n, "achieved": round(s_ach, 2), "frac": round(s_ach / HBM_PEAK_GBS, 5),
                                 "note": "the same sweep as a synchronous call runs it (look-ahead off, one pair at a time on the main stream): narrow "
                                         "strips, 7 compute waves -- the shorter chain; round 3's and earlier lines quoted this figure"}
        agg_ms, agg_n = tm["sgbm_wta"]
        if agg_n:
            s_in = (agg_ms / 1e3) / agg_n
            roof["in_stream"] = {"launch_us": round(s_in * 1e6, 2), "launches": int(agg_n), "achieved": round(alg_bytes / s_in / 1e9, 2),
                                 "note": "the same launch inside the timed region, beside the other pairs' kernels: a latency under contention"}
        # whole-job view: SURVEY's algorithmic bytes per pair x pairs/s against the peak
        roof["aggregate"] = {"algorithmic_bytes_per_pair": survey_bytes,
                             "achieved": round(survey_bytes * value / world / 1e9, 2),
                             "frac": round(survey_bytes * value / world / 1e9 / HBM_PEAK_GBS, 5)}
        if tb is None:
            roof.update({"achieved": roof["aggregate"]["achieved"], "frac": roof["aggregate"]["frac"],
                         "condition": "no post-pass in this run: whole-job algorithmic bytes x pairs/s"})
        if per_pair:
            roof["aggregate"]["measured_traffic_bytes_per_pair"] = per_pair
            roof["aggregate"]["measured_traffic_gb_per_s"] = round(per_pair * value / world / 1e9, 1)
        if isinstance(valu_pp, (int, float)) and valu_pp > 0:
            # the other roof of this integer pipeline: a wave64 packed-integer / DPP instruction holds its SIMD for 4 cycles
            issue_peak = 1024 * SHADER_GHZ * 1e9 / 4.0            # wave-instructions per second the 1024 SIMDs can issue
            roof["aggregate"]["vector_issue"] = {
                "wave_instructions_per_pair": int(valu_pp), "per_second": round(valu_pp * value / world, 0),
                "peak_per_second": issue_peak, "frac": round(valu_pp * value / world / issue_peak, 4),
                "note": "SQ_INSTS_VALU of every kernel of a pair (committed profile, not measured in this run) x pairs/s against 1024 "
                        "SIMDs x %.1f GHz / 4 cycles per wave64 instruction: the job sits between its two roofs, closer to this one" % SHADER_GHZ}
        if copy_gbs is not None:
            ceil = max(copy_gbs.values())
            roof["copy_ceiling"] = {"unit": "GB/s", "bytes_counted": "read + written", **copy_gbs,
                                    "frac_of_peak": round(ceil / HBM_PEAK_GBS, 4)}
            if per_pair and ceil > 0:
                roof["aggregate"]["measured_traffic_frac_of_copy_ceiling"] = round(per_pair * value / world / 1e9 / ceil, 4)
        out = {
            "metric": "stereo frame-pairs/sec (1280x720)" if workload == "C2" else "stereo frame-pairs/sec (%dx%d)" % (c.w, c.h),
            "value": round(value, 3), "unit": "frame-pairs/s", "n_gpus": world, "steps": K, "warmup": W,
            "ms_per_step": round(1e3 * dt_max / K, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "int16", "data": "synthetic",
            "config": {"workload": "%s: stereo %dx%d corridor stream, SGBM D=%d %s + ORB %d + "
                                   "Hamming kNN/ratio + 3-D lookup + rigid clique + Umeyama"
                                   % (workload, c.w, c.h, sgbm["numDisparities"], "8-path (MODE_HH)" if workload == "C4" else "5-path (MODE_SGBM)",
                                      ODO_KW["nfeatures"]),
                       "odometer": {k: ODO_KW[k] for k in ("rigidity_threshold", "outlier_threshold", "match_threshold", "min_matches")},
                       "parallelism": "frame-sharded x%d, pose all-gather over %s" % (world, group.transport if world > 1 else "nothing (1 rank)"),
                       "inputs": "host numpy every step (PCIe-inclusive)" if args.from_host and not light else "resident in HBM",
                       "deviations": DEVIATIONS},
            "roofline": roof,
            "accepted_frames": int(np.sum(all_ok)), "frames": int(len(all_ok)),
            "sgbm_sweep_error": int(sweep_err),
            # `value` = the median of R cold-start windows of exactly K steps each (consecutive frames of the sequence, every
            # stream drained and all look-ahead work dropped before each clock starts); the samples, in order:
            "window_values": [round(K * world / d, 2) for d in dt_windows],
            "window_ms": [round(1e3 * d, 3) for d in dt_windows],
            # the shader clock (MHz) rank 0 read right after each window's clock had stopped (vo_shader_clock: one wave, 200 us):
            # a GPU that was idle before the run ramps its clock over the first windows
            "window_clock_mhz": clocks[:len(dt_windows)],
            "cores_per_rank": cores_per_rank(world),
        }
        if dt_default:
            out["default_odometer"] = {
                "value": round(K * world / dt_default, 3), "unit": "frame-pairs/s", "steps": K, "ms_per_step": round(1e3 * dt_default / K, 4),
                "odometer": {"rigidity_threshold": 0, "outlier_threshold": 0},
                "note": "one cold-start window of the same K steps with the reference's default odometer "
                        "(stereo_odometer.py:14-15): no clique filter, no outlier pass"}
        if dt_steady:
            sv = S_steady * world / dt_steady
            out["steady_state"] = {
                "value": round(sv, 3), "unit": "frame-pairs/s", "steps": S_steady, "ms_per_step": round(1e3 * dt_steady / S_steady, 4),
                "accepted_frames": int(steady_acc),
                "aggregate_frac": round(survey_bytes * sv / world / 1e9 / HBM_PEAK_GBS, 5),
                "note": "one pass of %d update() steps from a cold start (fill and drain included), inputs resident in HBM; the "
                        "sequence walks this rank's %d rendered frames forwards and backwards (consecutive frames either way)"
                        % (S_steady, n_unique)}
        if world > 1:
            out["rccl"] = group.describe()        # what the communicator itself reports (ranks, this rank), not a string we made up
        if tb is not None:
            out["stage_ms_per_pair_alone"] = {k: round(v[0] / max(nb + 1, 1), 4) for k, v in tb.items()}
        if tmix is not None:
            out["stage_ms_per_pair_in_mix"] = {k: round(v[0] / max(v[1], 1), 4) for k, v in tmix.items()}
            out["stage_mix_pass_pairs_per_s"] = round(mix_rate, 1)
        if from_host_rate is not None:
            out["from_host_pairs_per_s"] = round(from_host_rate, 2)    # PCIe-inclusive; never `value`
            out["from_host_window"] = "480 pairs after 8 untimed ones (host numpy arrays through StereoOdometer.run; 96 frames walked forwards and backwards)"
        if world > 1:
            out["shard_boundaries_inexact"] = sharding.boundary_report(all_ok, R * K, world)
        # trajectory error vs the analytic ground truth (information only)
        poses = sharding.compose(all_rel, all_ok)
        if world == 1:
            gt0 = np.linalg.inv(Corridor.gt_pose(first + W - 1))
            err = [np.linalg.norm(poses[i][:3, 3] - (gt0 @ Corridor.gt_pose(first + W + i))[:3, 3]) for i in range(R * K)]
            out["ate_vs_ground_truth_m"] = round(float(np.sqrt(np.mean(np.square(err)))), 5)
        if world == 1 and args.cpu_pairs > 0 and not light:
            out["cpu_baseline"], out["ate_vs_cpu_m"] = cpu_baseline(c, cam, sgbm, frames, W, min(args.cpu_pairs, K), odo_poses=poses)
    ctx.close()                                   # the next workload of this process gets the whole device
    return out


def bench_c5(args, group, device, K_steps, W):
    """BASELINE config 5 (no openVO counterpart): monocular 1920x1080, ORB 8000 keypoints per frame, ~8000 x 8000
    Hamming kNN-2 + ratio, 5000-hypothesis essential-matrix RANSAC -- one MonoOdometer.update per step, frames
    resident in HBM, one host synchronisation per pair.  Not HBM-bound: the roofline block prices the
    two op-counts against the vector-ALU peak instead."""
    from openvo_amd.mono import MonoOdometer
    from openvo_amd.synth import Corridor
    c = Corridor("C5")
    iters, nfeat = 5000, 8000
    Kmat = np.array([[c.f, 0, c.cx], [0, c.f, c.cy], [0, 0, 1.0]])
    solver = int(os.environ.get("VO_C5_SOLVER", "5"))
    odo = MonoOdometer(Kmat, (c.w, c.h), nfeatures=nfeat, ransac_iters=iters, device=device, solver=solver)
    first = group.rank * K_steps
    n_img = min(W + K_steps, 256)                      # beyond 256 frames the stream wraps around (1 GB of HBM); the wrap's pair is rejected
    frames = [c.pair(first + k)[0] for k in range(n_img)]
    odo.stage_frames(frames)
    ctx = odo._ctx
    order = [k % n_img for k in range(W + K_steps)]
    for k in order[:W]:
        odo.update(k)
    odo.reset_lookahead()        # nothing computed before the clock starts may be used inside the timed region
    ctx.synchronize(); group.barrier()
    nq_nt, resid, acc = 0, 0, 0
    t0 = time.perf_counter()
    for k in order[W:]:
        ok = odo.update(k)
        acc += bool(ok)
        if odo.last is not None:
            resid += iters * odo.last["matches"]
    ctx.synchronize()
    dt = time.perf_counter() - t0
    group.barrier()
    dt = group.all_reduce_max(dt)
    spec, spec_depth = dict(odo.speculation), odo.speculate
    # per-stage device times from an untimed pass of the same steps with HIP events on (event packets around every stage of
    # every stream cost throughput: they stay out of the timed region)
    n_ev = min(K_steps, 40)
    odo.restart()                # (the pass replays the timed frames: its first one is a first frame again)
    odo.speculate = 0            # one pair's chain at a time: a stage's event pair then brackets that stage's own kernels only
    ctx.synchronize()
    ctx.enable_timing(True)
    ctx.timings(reset=True)
    resid_ev = 0
    for k in order[W:W + n_ev]:
        odo.update(k)
        if odo.last is not None:
            resid_ev += iters * odo.last["matches"]
    ctx.synchronize()
    tm = ctx.timings(reset=True)
    ctx.enable_timing(False)
    out = None
    if group.rank == 0:
        n_kp = ctx.orb_slot_count(odo._ref[0], nfeat, 0)
        pair_dists = float(n_kp) * n_kp * n_ev          # ~ keypoints^2 Hamming distances (256 bit) per pair
        match_s, pose_s = tm["match"][0] / 1e3, tm["pose"][0] / 1e3
        knn_s = tm["knn"][0] / 1e3                       # the kNN kernel alone (the match stage also holds the ratio compaction)
        # ... and the same launch 20 times back to back between ONE pair of events (a pair of events around a single 25 us
        # launch measures its own packets as well): the duration the roofline is computed from
        knn_us = None
        try:
            if odo._ref is not None:
                knn_us = ctx.measure_knn(odo._ref[0], odo._ref[0], 20)
        except Exception:
            knn_us = None
        if knn_us:
            knn_s = knn_us * 1e-6 * n_ev
        # vector-ALU peak: 256 CUs x 4 SIMD x 32 lanes x 2.4 GHz = 7.86e13 32-bit lane-ops/s; one 256-bit Hamming distance is
        # 8 xor + 8 popcount-accumulate lane-ops, one Sampson residual ~ 30 float lane-ops
        lane_ops = 256 * 4 * 32 * 2.4e9
        ham_rate = pair_dists / knn_s if knn_s > 0 else 0.0
        # the distance table is an int8 contraction on the matrix cores (popcount(q ^ t) = |q| + |t| - 2 q.t, exact): 256
        # multiply-adds per pair against the dense int8 MFMA peak (1024 MAC / cycle / SIMD: MI355X_MICROARCH.md, matrix cores)
        mfma_i8_peak = 1024 * 2 * 1024 * 2.4e9
        res_rate = resid_ev / pose_s if pose_s > 0 else 0.0
        out = {"metric": "mono frame-pairs/sec (1920x1080)", "value": round(K_steps * group.world / dt, 3), "unit": "frame-pairs/s",
               "n_gpus": group.world, "steps": K_steps, "warmup": W, "ms_per_step": round(1e3 * dt / K_steps, 4), "higher_is_better": True,
               "scaling": "weak", "vs_baseline": None, "dtype": "u8/f32", "data": "synthetic",
               "config": {"workload": "C5: mono 1920x1080 corridor stream, ORB %d + Hamming kNN-2/ratio + %d-hypothesis essential-matrix RANSAC "
                                      "(%s minimal solver), one host sync per pair" % (nfeat, iters, "five-point" if solver == 5 else "eight-point"),
                          "parallelism": "frame-sharded x%d" % group.world, "inputs": "resident in HBM",
                          "oracle": "none in openVO (no RANSAC, no monocular path): parity vs the build's own CPU restatement only"},
               "roofline": {"bound": "mfma", "kernel": "k_bf_knn2 (Hamming kNN-2 as an exact int8 contraction: v_mfma_i32_16x16x64_i8)",
                            "achieved": round(ham_rate * 512 / 1e12, 2), "peak": round(mfma_i8_peak / 1e12, 1), "unit": "TOP/s",
                            "frac": round(ham_rate * 512 / mfma_i8_peak, 5), "traffic": None,
                            "knn_us_per_launch": round(1e6 * knn_s / max(n_ev, 1), 2),
                            "knn_us_per_launch_single_event_pair": round(1e3 * tm["knn"][0] / max(n_ev, 1), 2),
                            "knn_condition": "20 launches back to back between two HIP events on the main stream (the frame's %d descriptors against themselves)" % int(n_kp),
                            "hamming_pair_distances_per_s": round(ham_rate, 0), "residual_evaluations_per_s": round(res_rate, 0),
                            "residual_frac_of_valu_peak": round(res_rate * 30 / lane_ops, 5),
                            "stage_ms_per_pair": {k: round(v[0] / n_ev, 4) for k, v in tm.items() if v[0] > 0},
                            "stage_pass": "%d untimed steps with HIP events on, one pair step at a time" % n_ev},
               "pairs_in_flight": {"speculate": spec_depth, "orb_lookahead": odo.lookahead, "steps_begun_ahead": spec["begun"],
                                   "used": spec["used"], "voided": spec["void"]},
               "accepted_frames": int(acc), "frames": K_steps, "keypoints_per_frame": int(n_kp),
               "matches_per_pair": int(resid / iters / max(K_steps, 1))}
    odo.close()
    return out if group.rank == 0 else None


def cores_per_rank(world):
    """Host cores this rank may run on (its affinity mask) divided among the ranks of this node: each rank's loop needs
    about 0.1 ms of one core per step."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    local = int(os.environ.get("LOCAL_WORLD_SIZE", world) or world)
    return round(n / max(local, 1), 2)


# ---- CPU baseline (reported, not the target) ------------------------------------------------------------------
def _port_chunk(c, cam, sgbm, frames, lo, n, dense=True):
    """The oracle odometer (scalar C port of the reference's OpenCV path) over frames[lo-1 : lo+n], the first
    being the halo; returns the c_T_w after each of the n frames (relative to the halo frame)."""
    from oracle.odometer import RefStereoCamera, RefStereoOdometer
    rcam = RefStereoCamera(cam.Q, cam.valid_region_left, sgbm, mode=int(sgbm.get("mode", 0)), dense_3d=dense)
    rodo = RefStereoOdometer(rcam, **ODO_KW)
    rodo.update(*frames[lo - 1])
    base = np.linalg.inv(rodo.c_T_w)
    out = []
    for i in range(n):
        rodo.update(*frames[lo + i])
        out.append(np.linalg.inv(rodo.c_T_w @ base))
    return out


class _cv2_seams:
    """Swap the oracle's ORB / matcher / Umeyama / Rodrigues for the real cv2 calls while the CPU baseline runs
    (the reference's call sites: stereo_odometer.py:117,163,190,204,212); cv2 objects are kept per thread."""

    def __init__(self, cv2):
        import threading
        self.cv2, self.tls = cv2, threading.local()

    def _objs(self):
        if not hasattr(self.tls, "orb"):
            self.tls.orb = self.cv2.ORB_create(nfeatures=ODO_KW["nfeatures"])
            self.tls.matcher = self.cv2.BFMatcher.create(self.cv2.NORM_HAMMING)
        return self.tls.orb, self.tls.matcher

    def __enter__(self):
        from oracle import oracle as O
        cv2 = self.cv2
        self.saved = (O.orb_detect_and_compute, O.bf_knn2_hamming, O.umeyama, O.rodrigues)

        def orb(img, mask, nfeatures, **kw):
            kps, desc = self._objs()[0].detectAndCompute(np.ascontiguousarray(img), mask)
            xy = np.array([k.pt for k in kps], np.float32).reshape(-1, 2)
            return {"xy": xy, "desc": desc if desc is not None else np.zeros((0, 32), np.uint8)}

        def knn(q, t):
            m = self._objs()[1].knnMatch(q, t, k=2)
            idx = np.array([[a.trainIdx, b.trainIdx] for a, b in m], np.int32).reshape(-1, 2)
            dist = np.array([[a.distance, b.distance] for a, b in m], np.int32).reshape(-1, 2)
            return idx, dist

        O.orb_detect_and_compute, O.bf_knn2_hamming = orb, knn
        O.umeyama = lambda src, dst, force_rotation=True: cv2.estimateAffine3D(src, dst, force_rotation=force_rotation)
        O.rodrigues = lambda R: cv2.Rodrigues(R)[0]
        return self

    def __exit__(self, *exc):
        from oracle import oracle as O
        O.orb_detect_and_compute, O.bf_knn2_hamming, O.umeyama, O.rodrigues = self.saved


def _cv2_chunk(cv2, c, cam, sgbm, frames, lo, n):
    """The reference's call sequence on a real cv2 (the build's own harness, SURVEY table 2.3): StereoSGBM.compute
    -> reprojectImageTo3D -> mask -> ORB.detectAndCompute -> BFMatcher.knnMatch(k=2) + ratio -> bilinear 3-D
    lookup -> clique -> estimateAffine3D(force_rotation=True) -> outlier pass -> gates -> chain.  The glue
    (state machine, ratio test, bilinear lookup, clique) is the oracle odometer's, which restates the reference's
    own numpy code; must run inside `_cv2_seams`."""
    from oracle import oracle as O
    from oracle.odometer import RefStereoOdometer

    class Dense:
        def __init__(self, a):
            self.a = np.ascontiguousarray(a)

        def sample(self, xy):
            return O.bilinear_at(self.a, xy)

    class Cv2Camera:
        def __init__(self):
            self.Q, self.valid_region_left = cam.Q, cam.valid_region_left
            self.m = cv2.StereoSGBM_create(sgbm["minDisparity"], sgbm["numDisparities"], sgbm["blockSize"], sgbm["P1"], sgbm["P2"],
                                           sgbm["disp12MaxDiff"], sgbm["preFilterCap"], sgbm["uniquenessRatio"],
                                           sgbm["speckleWindowSize"], sgbm["speckleRange"], mode=int(sgbm.get("mode", 0)))

        def compute_3d(self, L, R, preprocessed=False):
            disp = self.m.compute(L, R).astype(np.float32) / 16
            x3 = cv2.reprojectImageTo3D(disp, self.Q)
            vr = self.valid_region_left
            return Dense(x3[vr[1]:vr[3], vr[0]:vr[2]]), disp[vr[1]:vr[3], vr[0]:vr[2]], L[vr[1]:vr[3], vr[0]:vr[2]]

    rodo = RefStereoOdometer(Cv2Camera(), **ODO_KW)
    rodo.update(*frames[lo - 1])
    base = np.linalg.inv(rodo.c_T_w)
    out = []
    for i in range(n):
        rodo.update(*frames[lo + i])
        out.append(np.linalg.inv(rodo.c_T_w @ base))
    return out


def cpu_baseline(c, cam, sgbm, frames, W, n_pairs, odo_poses):
    """The CPU path on the first timed pairs of the same sequence: one thread (n_pairs pairs in order) and all
    host cores.  OpenCV's default-mode StereoSGBM is a single sequential raster pass per frame, so the way a CPU
    user fills the cores is the way this build fills GPUs: contiguous chunks of frames per worker, each with a
    halo frame.  Also returns the ATE between the GPU path and the one-thread CPU path on those pairs."""
    from concurrent.futures import ThreadPoolExecutor
    try:
        import cv2
    except Exception:
        cv2 = None
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    kind = "reference" if cv2 is not None else "port"
    if cv2 is not None:
        cv2.setNumThreads(1)
        run = lambda lo, n: _cv2_chunk(cv2, c, cam, sgbm, frames, lo, n)
    else:
        run = lambda lo, n: _port_chunk(c, cam, sgbm, frames, lo, n)
    import contextlib
    seams = _cv2_seams(cv2) if cv2 is not None else contextlib.nullcontext()
    with seams:
        return _cpu_legs(c, cv2, cores, kind, run, frames, W, n_pairs, odo_poses)


def _cpu_legs(c, cv2, cores, kind, run, frames, W, n_pairs, odo_poses):
    from concurrent.futures import ThreadPoolExecutor
    t0 = time.perf_counter()
    poses = run(W, n_pairs)
    dt1 = time.perf_counter() - t0
    err = [np.linalg.norm(poses[i][:3, 3] - odo_poses[i][:3, 3]) for i in range(n_pairs)]
    ate = float(np.sqrt(np.mean(np.square(err))))
    # all cores: `cores` workers (the C code and cv2 release the GIL), two pairs each plus the halo frame
    per = 2
    n_all = min(cores * per, len(frames) - W)
    workers = max(1, n_all // per)
    if cv2 is not None:
        cv2.setNumThreads(1)      # one frame per core; OpenCV's inner threading stays off so the cores are not oversubscribed
    t0 = time.perf_counter()
    with ThreadPoolExecutor(max_workers=workers) as ex:
        list(ex.map(lambda k: run(W + k * per, per), range(workers)))
    dta = time.perf_counter() - t0
    n_done = workers * per
    what = ("cv2 %s (the reference's call sequence issued by the build's own harness)" % cv2.__version__) if cv2 is not None else \
        "scalar C port of the reference's OpenCV path (oracle/); OpenCV is not importable on this box"
    return ({"value": round(n_done / dta, 4), "unit": "frame-pairs/s", "cores": workers, "kind": kind,
             "sample": "%d pairs of the same %s sequence on %d worker threads (contiguous 2-pair chunks + halo frame, %.1f s); %s; "
                       "whole update() incl. the full reprojectImageTo3D" % (n_done, c.name, workers, dta, what),
             "one_thread": {"value": round(n_pairs / dt1, 4), "cores": 1,
                            "sample": "first %d timed pairs in order, %.1f s" % (n_pairs, dt1)}}, round(ate, 9))


if __name__ == "__main__":
    main()
