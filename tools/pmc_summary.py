"""Reduce two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate runs, as MI355X_MICROARCH.md's
HBM section prescribes) to profiles/<tag>_pmc_<wl>_summary.csv and profiles/traffic_<WL>.json.

usage: pmc_summary.py FETCH_counter_collection.csv WRITE_counter_collection.csv TAG WORKLOAD
gfx950 correction: FETCH_SIZE counts wide (16 B/lane) coalesced reads at half their size -> x2 for the
volume kernels, which only issue dwordx4 loads; WRITE_SIZE is exact.  Units: KB (1024 B)."""
import collections, csv, json, os, sys

def load(path):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        acc[(r["Counter_Name"], r["Kernel_Name"].split("(")[0])].append(float(r["Counter_Value"]))
    return acc

fetch, write, tag, wl = sys.argv[1:5]
root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles")
rows = []
for acc in (load(fetch), load(write)):
    for (cn, kn), v in acc.items():
        rows.append((cn, kn, len(v), sum(v) / len(v)))
rows.sort(key=lambda r: (r[0], -r[3]))
with open(os.path.join(root, "%s_pmc_%s_summary.csv" % (tag, wl.lower())), "w") as f:
    f.write("counter,kernel,dispatches,mean_value_KB_raw\n")
    for r in rows:
        if r[3] >= 64.0:
            f.write("%s,%s,%d,%.1f\n" % r)
def get(cn, prefix):
    for r in rows:
        if r[0] == cn and prefix in r[1]:
            return r[1], r[2], r[3]
    return None, 0, 0.0
kn, nd, fk = get("FETCH_SIZE", "k_sgbm_paths")
_, _, wk = get("WRITE_SIZE", "k_sgbm_paths")
out = {"workload": wl, "kernel": kn, "dispatches": nd,
       "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, two separate passes of `python bench.py --steps 8 --warmup 2 --cpu-pairs 0`, MI355X",
       "FETCH_SIZE_KB_raw": round(fk), "WRITE_SIZE_KB_raw": round(wk),
       "correction": "gfx950: FETCH_SIZE counts wide coalesced (16 B/lane) reads at half -> x2 (MI355X_MICROARCH.md, HBM section); WRITE_SIZE exact",
       "sgbm_path_bytes_per_launch": int(round((2 * fk + wk) * 1024)), "other_kernels_KB_raw": {}}
for pre in ("k_sgbm_vwta", "k_sgbm_cost_sweep", "k_sgbm_wta", "k_ccl_vmerge"):
    k2, _, f2 = get("FETCH_SIZE", pre)
    _, _, w2 = get("WRITE_SIZE", pre)
    if k2:
        out["other_kernels_KB_raw"][k2] = {"FETCH_SIZE": round(f2), "WRITE_SIZE": round(w2)}
json.dump(out, open(os.path.join(root, "traffic_%s.json" % wl), "w"), indent=1)
print(json.dumps(out, indent=1))
