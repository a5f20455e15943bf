#!/bin/bash
# Per-kernel wave / instruction counters of the bench loop, quickly (via gpurun): tools/pmc_quick.sh TAG [kernel-name-filter]
tag=$1; filt=${2:-k_}
out=$GRAFT_REPO_ROOT/gpurun_out/pmcq_$tag
mkdir -p $out
export TMPDIR=/tmp GPU_MAX_HW_QUEUES=24
cd /tmp
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU --kernel-trace --output-format csv -d $out/occ -o pmc -- python3 $GRAFT_REPO_ROOT/bench.py --cpu-pairs 0 --no-post --no-other --repeats 1 --steps 6 --warmup 2 > /dev/null 2> $out/err.txt || { tail -5 $out/err.txt; exit 1; }
python3 - <<PY
import csv, glob, collections
f = glob.glob("$out/occ/**/*counter_collection.csv", recursive=True)[0]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
seen = set()
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"].split("(")[0].replace("void ", "")
    if "$filt" not in k: continue
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    key = (k, r["Dispatch_Id"])
    if key not in seen: seen.add(key); n[k] += 1
print("%-40s %5s %9s %12s %12s %10s %10s" % ("kernel", "disp", "waves", "valu/wave", "cycles/wave", "salu/wave", "lds/wave"))
tot = 0
for k, c in sorted(acc.items(), key=lambda kv: -kv[1]["SQ_INSTS_VALU"]):
    w = c["SQ_WAVES"] or 1
    print("%-40s %5d %9.0f %12.0f %12.0f %10.0f %10.0f   valu per launch %.2f M" % (k[:40], n[k], w / n[k], c["SQ_INSTS_VALU"] / w, c["SQ_WAVE_CYCLES"] / w, c["SQ_INSTS_SALU"] / w, c["SQ_INSTS_LDS"] / w, c["SQ_INSTS_VALU"] / n[k] / 1e6))
    tot += c["SQ_INSTS_VALU"] / n[k]
print("sum over kernels of VALU wave-instructions per launch: %.1f M" % (tot / 1e6))
PY
rm -rf $out/occ
