#!/usr/bin/env python3
"""bench.py -- stereo frame-pairs/sec of the openVO hot path on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W
    (N > 1: launched by `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...`)

One step = one StereoOdometer.update() on one 1280x720 stereo pair of the synthetic corridor
sequence (BASELINE config 2, "C2": SGBM D=128 5-path + ORB 500 + Hamming kNN/ratio + 3-D lookup +
rigid-clique filter + Umeyama), inputs already resident in HBM.  Each rank owns one GPU and a
contiguous chunk of the sequence (weak scaling); the only exchange is the final all_gather of
the relative poses (RCCL).  Rank 0 prints ONE JSON line with the contract fields plus
`roofline` (dominant kernel: one SGBM aggregation path, timed with HIP events on the library's
own stream) and `cpu_baseline` (the CPU oracle timed on this box's host cores, N = 1 only).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

os.environ.setdefault("GPU_MAX_HW_QUEUES", "10")   # before any HIP runtime starts (see openvo_amd/__init__.py)

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

ODO_KW = dict(nfeatures=500, match_threshold=0.8, rigidity_threshold=0.1, outlier_threshold=0.02,
              preprocessed_frames=True, min_matches=10)
HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak (MI355X_MICROARCH.md)


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
    ap.add_argument("--cpu-pairs", type=int, default=4, help="pairs the CPU oracle is timed on (0 = skip)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    dist = None
    torch = None
    use_cuda = True
    ndev = 0
    if world > 1:
        import torch
        import torch.distributed as dist
        ndev = torch.cuda.device_count()
        # one GPU per rank -> RCCL ("nccl"); fewer GPUs than ranks (rehearsal on a 1-GPU box) -> ranks
        # share devices and the tiny pose gather goes over gloo
        use_cuda = ndev >= world
        if use_cuda:
            torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl" if use_cuda else "gloo", rank=rank, world_size=world)
    else:
        try:
            import torch
        except Exception:  # torch is plumbing only (barrier/sync); the path itself does not need it
            torch = None
    if args.gpus != world and rank == 0 and world > 1:
        print("warning: --gpus %d but WORLD_SIZE %d" % (args.gpus, world), file=sys.stderr)

    from openvo_amd import StereoCamera, StereoOdometer, sharding
    from openvo_amd.synth import Corridor

    K, W = args.steps, args.warmup
    c = Corridor(args.workload)
    # config 4 is the 8-path (MODE_HH) cost-volume stress case; every other workload runs the reference's 5-path default
    sgbm = c.sgbm_params(mode=1) if args.workload == "C4" else c.sgbm_params()
    if args.ndisp:
        sgbm["numDisparities"] = args.ndisp
    cam = StereoCamera(c.K(), c.dist(), c.K(), c.dist(), c.rect_params(), sgbm, (c.w, c.h),
                       device=(local_rank if (world == 1 or use_cuda) else local_rank % max(ndev, 1)),
                       max_keypoints=ODO_KW["nfeatures"])
    odo = StereoOdometer(cam, **ODO_KW)
    # this rank's frames: W warm-up frames (they also provide the halo) then K timed frames
    first = rank * K
    frames = c.pairs(first, W + K)
    staged = cam.stage_pairs(frames)          # inputs resident in HBM before the clock starts
    ctx = cam._ctx

    def sync_all():
        ctx.synchronize()
        if torch is not None and torch.cuda.is_available():
            torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()

    # the interpreter's cyclic GC (a full collection with torch loaded costs ~50 ms) must not fire
    # inside the timed region: collect now, then keep it off until the clock stops
    import gc
    gc.collect()
    gc.freeze()
    gc.disable()
    for i in range(W):
        odo.update(staged[i], None)
    # HIP events (recorded on the library's stream, resolved after the run) around the dominant
    # kernel only: event packets around every small stage would perturb the throughput measured
    ctx.enable_timing(not args.no_events, stages=["sgbm_agg"])
    ctx.timings(reset=True)
    rel, acc = [], []
    cam.reset_lookahead()      # nothing computed before the clock starts may be used inside the timed region
    sync_all()
    t0 = time.perf_counter()
    if args.from_host:
        before = odo.c_T_w
        for ok in odo.run(frames[W:W + K]):
            acc.append(bool(ok))
            rel.append(sharding.relative_from_chain(before, odo.c_T_w) if ok else np.eye(4))
            before = odo.c_T_w
    else:
        for i in range(W, W + K):
            before = odo.c_T_w
            ok = odo.update(staged[i], None)
            acc.append(bool(ok))
            rel.append(sharding.relative_from_chain(before, odo.c_T_w) if ok else np.eye(4))
    sync_all()
    dt = time.perf_counter() - t0
    gc.enable()
    tm = ctx.timings(reset=True)
    # per-stage breakdown (information only): a short untimed post-pass with every stage timed and the
    # look-ahead engines off, i.e. one pair at a time with each kernel alone on the GPU -- the same
    # condition a rocprofv3 kernel trace imposes (it serialises dispatches)
    ctx.enable_timing(True)
    nb = min(8, K)
    la = cam.lookahead
    cam.reset_lookahead()
    cam.lookahead = 0
    probe = StereoOdometer(cam, **ODO_KW)
    for i in range(W + K - nb - 1, W + K):
        probe.update(staged[i], None)
    tb = ctx.timings(reset=True)
    ctx.enable_timing(False)
    cam.lookahead = la

    # max over ranks of the timed region
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda" if use_cuda else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt_max = float(t.item())
    else:
        dt_max = dt
    # final pose gather (the path's only exchange): 16 float64 + accept flag per frame
    dev = ("cuda:%d" % local_rank) if (dist is not None and use_cuda) else None
    all_rel, all_ok = sharding.gather_relative(np.array(rel), np.array(acc, np.float64), dist, dev)

    if rank == 0:
        total_pairs = K * world
        value = total_pairs / dt_max
        cells, npaths = ctx.sgbm_last_geometry()
        agg_ms, agg_n = tm["sgbm_agg"]
        n_launch = agg_n                                               # one k_sgbm_paths launch per pair
        per_launch_s = (agg_ms / 1e3) / max(n_launch, 1)
        alg_bytes = 2.0 * cells * npaths                               # the int16 cost volume read once per path direction
        achieved = alg_bytes / per_launch_s / 1e9 if per_launch_s > 0 else 0.0
        iso_ms, iso_n = tb["sgbm_agg"]
        iso_s = (iso_ms / 1e3) / max(iso_n, 1)
        iso_ach = alg_bytes / iso_s / 1e9 if iso_s > 0 else 0.0
        traffic = None
        tf = os.path.join(ROOT, "profiles", "traffic_%s.json" % args.workload)
        if os.path.exists(tf):
            try:
                traffic = json.load(open(tf)).get("sgbm_path_bytes_per_launch")
            except Exception:
                traffic = None
        out = {
            "metric": "stereo frame-pairs/sec (1280x720)" if args.workload == "C2" else "stereo frame-pairs/sec (%dx%d)" % (c.w, c.h),
            "value": round(value, 3), "unit": "frame-pairs/s", "n_gpus": world, "steps": K, "warmup": W,
            "ms_per_step": round(1e3 * dt_max / K, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "int16", "data": "synthetic",
            "config": {"workload": "%s: stereo %dx%d corridor stream, SGBM D=%d %s + ORB %d + "
                                   "Hamming kNN/ratio + 3-D lookup + rigid clique + Umeyama"
                                   % (args.workload, c.w, c.h, c.D, "8-path (MODE_HH)" if args.workload == "C4" else "5-path (MODE_SGBM)",
                                      ODO_KW["nfeatures"]),
                       "odometer": {k: ODO_KW[k] for k in ("rigidity_threshold", "outlier_threshold", "match_threshold", "min_matches")},
                       "parallelism": "frame-sharded x%d, pose all_gather" % world,
                       "inputs": "host numpy every step (PCIe-inclusive)" if args.from_host else "resident in HBM"},
            "roofline": {"bound": "hbm", "kernel": "k_sgbm_paths (%d aggregation directions in one launch; the last one runs fused with the WTA)" % npaths,
                         "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic,
                         "bytes_per_launch": alg_bytes, "launch_us": round(per_launch_s * 1e6, 2), "launches": n_launch,
                         # the timed region overlaps several pairs, so a launch shares HBM with other kernels;
                         # alone on the GPU (post-pass, = what a serialising kernel trace reports) it takes:
                         "alone": {"launch_us": round(iso_s * 1e6, 2), "achieved": round(iso_ach, 2),
                                   "frac": round(iso_ach / HBM_PEAK_GBS, 5)}},
            "stage_ms_per_pair_alone": {k: round(v[0] / max(nb + 1, 1), 4) for k, v in tb.items()},
            "accepted_frames": int(np.sum(all_ok)), "frames": int(len(all_ok)),
        }
        # trajectory error vs the analytic ground truth (information only)
        poses = sharding.compose(all_rel, all_ok)
        if world == 1:
            gt0 = np.linalg.inv(Corridor.gt_pose(first + W - 1))
            err = [np.linalg.norm(poses[i][:3, 3] - (gt0 @ Corridor.gt_pose(first + W + i))[:3, 3]) for i in range(K)]
            out["ate_vs_ground_truth_m"] = round(float(np.sqrt(np.mean(np.square(err)))), 5)
        if world == 1 and args.cpu_pairs > 0:
            out["cpu_baseline"], out["ate_vs_cpu_m"] = cpu_baseline(c, cam, frames, W, args.cpu_pairs, odo_poses=poses)
        print(json.dumps(out))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def cpu_baseline(c, cam, frames, W, n_pairs, odo_poses):
    """The CPU oracle (a scalar C port of the reference's OpenCV path, 1 thread) on the first
    n_pairs timed pairs of the same sequence, preceded by the halo frame so that every timed pair
    yields a pose.  Also returns the ATE between the GPU path and this CPU path on those pairs."""
    from oracle.odometer import RefStereoCamera, RefStereoOdometer
    rcam = RefStereoCamera(cam.Q, cam.valid_region_left, c.sgbm_params(mode=1) if c.name == "C4" else c.sgbm_params())
    kw = dict(ODO_KW)
    rodo = RefStereoOdometer(rcam, **kw)
    rodo.update(*frames[W - 1])               # untimed: establishes `current`
    base = np.linalg.inv(rodo.c_T_w)
    t0 = time.perf_counter()
    poses = []
    for i in range(n_pairs):
        rodo.update(*frames[W + i])
        poses.append(np.linalg.inv(rodo.c_T_w @ base))
    dt = time.perf_counter() - t0
    err = [np.linalg.norm(poses[i][:3, 3] - odo_poses[i][:3, 3]) for i in range(n_pairs)]
    ate = float(np.sqrt(np.mean(np.square(err))))
    return ({"value": round(n_pairs / dt, 4), "unit": "frame-pairs/s", "cores": 1, "kind": "port",
             "sample": "first %d timed pairs of the same %s sequence (%.1f s of CPU work), single thread; "
                       "OpenCV itself is not installed on this box" % (n_pairs, c.name, dt)}, round(ate, 9))


if __name__ == "__main__":
    main()
