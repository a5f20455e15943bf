"""Reduce gpurun_out/prof_<tag>/ (tools/profile_round2.sh) to the committed artefacts under profiles/:
  <tag>_kernel_stats_c2.csv   rocprofv3 --kernel-trace --stats of the bench command
  <tag>_pmc_c2_summary.csv    FETCH_SIZE / WRITE_SIZE per kernel, line scheme and raster scheme
  <tag>_occupancy_c2.csv      waves, waves per SIMD, VALU-busy and stall shares of the SGBM volume kernels
  traffic_C2.json             per-launch bytes of the bench line's roofline kernel + per-PAIR SGBM totals of both schemes
gfx950 correction: FETCH_SIZE counts wide (16 B/lane) coalesced reads at half their size -> x2 (MI355X_MICROARCH.md,
HBM section); WRITE_SIZE is exact.  Units of the raw counters: KB (1024 B)."""
import collections, csv, json, os, shutil, sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "gpurun_out", "prof_%s" % tag)
dst = os.path.join(root, "profiles")
shutil.copy(os.path.join(src, "stats", "bench_kernel_stats.csv"), os.path.join(dst, "%s_kernel_stats_c2.csv" % tag))
shutil.copy(os.path.join(src, "bench_under_rocprof.json"), os.path.join(dst, "%s_bench_under_rocprof_c2.json" % tag))


def load(path):
    acc = collections.defaultdict(list)
    meta = {}
    for r in csv.DictReader(open(path)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        acc[(r["Counter_Name"], k)].append(float(r["Counter_Value"]))
        meta[k] = (int(r["Grid_Size"]), int(r["Workgroup_Size"]), int(r["VGPR_Count"]), int(r["LDS_Block_Size"]))
    return acc, meta


SGBM = ("k_sgbm_planes", "k_sgbm_cost_sweep", "k_sgbm_paths", "k_sgbm_we", "k_sgbm_vwta", "k_sgbm_raster", "k_sgbm_fin", "k_lr_median3", "k_ccl_")
rows, per_pair = [], {}
for scheme, name in ((0, "line"), (1, "raster"), (2, "line_we_fused")):
    f, _ = load(os.path.join(src, "fetch_%d" % scheme, "pmc_counter_collection.csv"))
    w, _ = load(os.path.join(src, "write_%d" % scheme, "pmc_counter_collection.csv"))
    tot_f = tot_w = 0.0
    kernels = sorted({k for (_, k) in list(f) + list(w)})
    # dispatches per pair: the SGBM kernels run once per pair (k_sgbm_raster: once, or twice for MODE_HH)
    npairs = len(f.get(("FETCH_SIZE", next(k for k in kernels if k.startswith("k_sgbm_planes"))), [1]))
    for k in kernels:
        fv, wv = f.get(("FETCH_SIZE", k), []), w.get(("WRITE_SIZE", k), [])
        mf, mw = (sum(fv) / len(fv) if fv else 0.0), (sum(wv) / len(wv) if wv else 0.0)
        if mf + mw >= 64:
            rows.append((name, k, max(len(fv), len(wv)), mf, mw))
        if k.startswith(SGBM):
            tot_f += sum(fv) / npairs
            tot_w += sum(wv) / npairs
    per_pair[name] = {"FETCH_SIZE_KB_raw": round(tot_f), "WRITE_SIZE_KB_raw": round(tot_w),
                      "bytes_per_pair_corrected": int(round((2 * tot_f + tot_w) * 1024)), "pairs_profiled": npairs}
rows.sort(key=lambda r: (r[0], -(r[3] * 2 + r[4])))
with open(os.path.join(dst, "%s_pmc_c2_summary.csv" % tag), "w") as fh:
    fh.write("scheme,kernel,dispatches,FETCH_SIZE_mean_KB_raw,WRITE_SIZE_mean_KB_raw,bytes_per_launch_corrected\n")
    for r in rows:
        fh.write("%s,%s,%d,%.1f,%.1f,%d\n" % (r[0], r[1], r[2], r[3], r[4], int((2 * r[3] + r[4]) * 1024)))

# occupancy / issue / stall shares
with open(os.path.join(dst, "%s_occupancy_c2.csv" % tag), "w") as fh:
    fh.write("scheme,kernel,dispatches,grid_threads,workgroup,vgprs,lds_bytes,waves,waves_per_simd_if_all_resident,"
             "valu_busy_frac_of_wave_cycles,active_any_frac,wait_any_frac(parked: s_waitcnt/barrier),wait_inst_any_frac(issue stall),"
             "wave_cycles_per_wave,gui_active_cycles\n")
    for scheme, name in ((0, "line"), (1, "raster"), (2, "line_we_fused")):
        o, meta = load(os.path.join(src, "occ_%d" % scheme, "pmc_counter_collection.csv"))
        for k in sorted(meta):
            if not k.startswith(("k_sgbm_cost_sweep", "k_sgbm_paths", "k_sgbm_we", "k_sgbm_vwta", "k_sgbm_raster", "k_orb_select", "k_pose_solve", "k_orb_describe", "k_orb_fast")):
                continue
            g = lambda c: (sum(o[(c, k)]) / len(o[(c, k)])) if o.get((c, k)) else 0.0
            waves, wc = g("SQ_WAVES"), g("SQ_WAVE_CYCLES")
            grid, wg, vg, lds = meta[k]
            fh.write("%s,%s,%d,%d,%d,%d,%d,%.0f,%.2f,%.3f,%.3f,%.3f,%.3f,%.0f,%.0f\n" % (
                name, k, len(o[("SQ_WAVES", k)]), grid, wg, vg, lds, waves, waves / 1024.0,
                g("SQ_ACTIVE_INST_VALU") / wc if wc else 0, g("SQ_ACTIVE_INST_ANY") / wc if wc else 0, g("SQ_WAIT_ANY") / wc if wc else 0,
                g("SQ_WAIT_INST_ANY") / wc if wc else 0, wc / waves if waves else 0, g("GRBM_GUI_ACTIVE")))

pk = [r for r in rows if r[0] == "line" and r[1].startswith("k_sgbm_paths")]
out = {"workload": "C2", "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate passes of `python bench.py --steps 6 --warmup 2 "
                                   "--cpu-pairs 0 --no-post`, MI355X; line scheme with separate W / E volumes (VO_WE_FUSE=0), raster scheme (VO_RASTER=1), "
                                   "line scheme with W + E stored as one volume (VO_WE_FUSE=1: what the default per-pair policy picks for pairs behind a queue)",
       "correction": "gfx950: FETCH_SIZE counts wide coalesced (16 B/lane) reads at half -> x2 (MI355X_MICROARCH.md, HBM section); WRITE_SIZE exact",
       "kernel": pk[0][1] if pk else None,
       "sgbm_path_bytes_per_launch": int((2 * pk[0][3] + pk[0][4]) * 1024) if pk else None,
       "sgbm_bytes_per_pair": per_pair}
json.dump(out, open(os.path.join(dst, "traffic_C2.json"), "w"), indent=1)
print(json.dumps(out, indent=1))
print(open(os.path.join(dst, "%s_occupancy_c2.csv" % tag)).read())
