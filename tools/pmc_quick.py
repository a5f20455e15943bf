"""Per-kernel HBM traffic of one profile set (tools/profile_round3.sh): python tools/pmc_quick.py gpurun_out/prof_<tag>"""
import collections, csv, glob, os, sys

src = sys.argv[1]


def load(sub, counter):
    acc = collections.defaultdict(list)
    for path in glob.glob(os.path.join(src, sub, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(path)):
            if r["Counter_Name"] == counter:
                acc[r["Kernel_Name"].split("(")[0].replace("void ", "")].append(float(r["Counter_Value"]))
    return acc


f, w = load("fetch", "FETCH_SIZE"), load("write", "WRITE_SIZE")
npairs = max(len(v) for k, v in f.items() if k.startswith("k_sgbm_planes"))
tot = 0.0
print("pairs profiled:", npairs)
for k in sorted(set(f) | set(w), key=lambda k: -(2 * sum(f.get(k, [])) + sum(w.get(k, [])))):
    b = (2 * sum(f.get(k, [])) + sum(w.get(k, []))) * 1024 / npairs
    if b > 1e6:
        print("%-70s launches/pair %.2f  fetch_raw_KB %.0f write_KB %.0f  bytes/pair (FETCH x2 + WRITE) %.1f MB" % (
            k[:70], max(len(f.get(k, [])), len(w.get(k, []))) / npairs, sum(f.get(k, [])) / npairs, sum(w.get(k, [])) / npairs, b / 1e6))
    if k.startswith(("k_sgbm", "k_lr_", "k_ccl")):
        tot += b
print("SGBM bytes per pair: %.3f GB" % (tot / 1e9))
