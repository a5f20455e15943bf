"""Reduce gpurun_out/prof_<tag>/ (tools/profile_round.sh) to the committed artefacts under profiles/:
  <tag>_kernel_stats_c2.csv      rocprofv3 --kernel-trace --stats of the bench command (16 pairs in flight)
  <tag>_kernel_stats_alone_c2.csv  the same, one pair at a time (look-ahead off): every kernel alone on the GPU
  <tag>_bench_under_rocprof_c2.json
  <tag>_pmc_c2_summary.csv       FETCH_SIZE / WRITE_SIZE per kernel (separate passes)
  <tag>_occupancy_c2.csv         waves, VALU-busy and stall shares of the SGBM kernels (when the occ pass was run)
  traffic_C2.json                per-launch bytes of the roofline kernel + SGBM bytes per pair (read by bench.py)
gfx950 correction: FETCH_SIZE counts wide (16 B/lane) coalesced reads at half their size -> x2 (MI355X_MICROARCH.md,
HBM section); WRITE_SIZE is exact.  Units of the raw counters: KB (1024 B).
usage: python tools/pmc_summary.py <tag> [workload]"""
import collections, csv, glob, json, os, shutil, sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r04"
wl = sys.argv[2] if len(sys.argv) > 2 else "C2"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "gpurun_out", "prof_%s" % tag)
dst = os.path.join(root, "profiles")
low = wl.lower()
digest = None
if os.path.exists(os.path.join(src, "csrc_digest.txt")):
    digest = open(os.path.join(src, "csrc_digest.txt")).read().strip() or None
for pat, name in (("stats/**/*kernel_stats.csv", "%s_kernel_stats_%s.csv" % (tag, low)),
                  ("alone/**/*kernel_stats.csv", "%s_kernel_stats_alone_%s.csv" % (tag, low)),
                  ("alone_engine/**/*kernel_stats.csv", "%s_kernel_stats_alone_engine_%s.csv" % (tag, low))):
    f = glob.glob(os.path.join(src, pat), recursive=True)
    if f:
        shutil.copy(f[0], os.path.join(dst, name))
if os.path.exists(os.path.join(src, "bench_under_rocprof.json")):
    shutil.copy(os.path.join(src, "bench_under_rocprof.json"), os.path.join(dst, "%s_bench_under_rocprof_%s.json" % (tag, low)))


def load(sub):
    acc, meta = collections.defaultdict(list), {}
    for path in glob.glob(os.path.join(src, sub, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(path)):
            k = r["Kernel_Name"].split("(")[0].replace("void ", "")
            acc[(r["Counter_Name"], k)].append(float(r["Counter_Value"]))
            meta[k] = (int(r["Grid_Size"]), int(r["Workgroup_Size"]), int(r["VGPR_Count"]), int(r["LDS_Block_Size"]))
    return acc, meta


SGBM = ("k_sgbm", "k_lr_", "k_ccl")
# (launches of one run can differ between the two passes by the few that straddle a profiler window: use what both hold)
f, _ = load("fetch")
w, _ = load("write")
kernels = sorted({k for (_, k) in list(f) + list(w)})
# launches of the kernel every pair (C2 / C4) or frame (C5: monocular, no SGBM) runs exactly once
_once = [len(v) for (c, k), v in f.items() if k.startswith("k_sgbm_planes")] or [len(v) for (c, k), v in f.items() if k.startswith("k_orb_pyramid")]
npairs = max(_once)
mono = not any(k.startswith("k_sgbm_planes") for (_, k) in f)
rows, tot_f, tot_w, dom = [], 0.0, 0.0, None
for k in kernels:
    fv, wv = f.get(("FETCH_SIZE", k), []), w.get(("WRITE_SIZE", k), [])
    mf, mw = (sum(fv) / len(fv) if fv else 0.0), (sum(wv) / len(wv) if wv else 0.0)
    if mf + mw >= 64:
        rows.append((k, max(len(fv), len(wv)), mf, mw))
    if k.startswith(SGBM) or (mono and k.startswith("k_")):      # (config 5: every kernel of the pipeline)
        tot_f += sum(fv) / npairs
        tot_w += sum(wv) / npairs
    if ((k.startswith("k_sgbm_diag") and ", false, true>" in k) or (mono and k.startswith("k_bf_knn2"))) and (dom is None or max(len(fv), len(wv)) > dom[2]):
        dom = (k, int((2 * mf + mw) * 1024), max(len(fv), len(wv)))    # (the strip width most of the pairs ran with)
rows.sort(key=lambda r: -(r[2] * 2 + r[3]))
with open(os.path.join(dst, "%s_pmc_%s_summary.csv" % (tag, low)), "w") as fh:
    fh.write("kernel,dispatches,FETCH_SIZE_mean_KB_raw,WRITE_SIZE_mean_KB_raw,bytes_per_launch_corrected\n")
    for r in rows:
        fh.write("\"%s\",%d,%.1f,%.1f,%d\n" % (r[0], r[1], r[2], r[3], int((2 * r[2] + r[3]) * 1024)))
per_pair = int(round((2 * tot_f + tot_w) * 1024))
o, meta = load("occ")
valu_per_pair = None
if o:
    # vector instructions of ALL kernels of a pair (SQ_INSTS_VALU counts wave-instructions), from the occupancy pass
    n_occ = max(len(v) for (c, k), v in o.items() if c == "SQ_INSTS_VALU" and k.startswith("k_orb_pyramid" if mono else ("k_sgbm_planes", "k_orb_pyramid")))
    valu_per_pair = int(sum(sum(v) for (c, k), v in o.items() if c == "SQ_INSTS_VALU" and k.startswith("k_")) / n_occ)
json.dump({"workload": wl, "csrc_digest": digest, "tag": tag,
           "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate passes of `python bench.py --steps 6 --warmup 2 --cpu-pairs 0 --no-post` "
                     "(tools/profile_round.sh %s), MI355X, default schedule (W + E volume, diagonal sweep)" % tag,
           "correction": "gfx950: FETCH_SIZE counts wide coalesced (16 B/lane) reads at half -> x2 (MI355X_MICROARCH.md, HBM section); WRITE_SIZE exact",
           "kernel": dom[0] if dom else None, "dominant_kernel_bytes_per_launch": dom[1] if dom else None,
           "sgbm_bytes_per_pair": per_pair, "FETCH_SIZE_KB_raw_per_pair": round(tot_f), "WRITE_SIZE_KB_raw_per_pair": round(tot_w),
           "pairs_profiled": npairs,
           "valu_wave_instructions_per_pair": valu_per_pair,
           "valu_source": "SQ_INSTS_VALU summed over every kernel of the pipeline (the occupancy pass of the same script), per pair"},
          open(os.path.join(dst, "traffic_%s.json" % wl), "w"), indent=1)
print("SGBM bytes per pair: %.3f GB over %d pairs; dominant kernel %s; vector wave-instructions per pair %s" % (per_pair / 1e9, npairs, dom, valu_per_pair))

if o:
    with open(os.path.join(dst, "%s_occupancy_%s.csv" % (tag, low)), "w") as fh:
        fh.write("kernel,dispatches,grid_threads,workgroup,vgprs,lds_bytes,waves,waves_per_simd_if_all_resident,valu_busy_frac_of_wave_cycles,"
                 "active_any_frac,wait_any_frac(parked: s_waitcnt/barrier),wait_inst_any_frac(issue stall),wave_cycles_per_wave,valu_insts_per_wave,gui_active_cycles\n")
        for k in sorted(meta):
            if not k.startswith(("k_sgbm", "k_orb", "k_pose", "k_bf", "k_lr", "k_ccl", "k_ransac", "k_ratio")):
                continue
            g = lambda c: (sum(o.get((c, k), [0.0])) / max(len(o.get((c, k), [0.0])), 1))
            waves, wc = g("SQ_WAVES"), g("SQ_WAVE_CYCLES")
            if waves <= 0 or wc <= 0:
                continue
            fh.write("\"%s\",%d,%d,%d,%d,%d,%d,%.2f,%.3f,%.3f,%.3f,%.3f,%d,%d,%d\n" % (
                k, len(o.get(("SQ_WAVES", k), [])), meta[k][0], meta[k][1], meta[k][2], meta[k][3], waves, waves / 1024.0,
                g("SQ_ACTIVE_INST_VALU") / wc, g("SQ_ACTIVE_INST_ANY") / wc, g("SQ_WAIT_ANY") / wc, g("SQ_WAIT_INST_ANY") / wc,
                wc / waves, g("SQ_INSTS_VALU") / waves, g("GRBM_GUI_ACTIVE")))
    print("occupancy table written")
