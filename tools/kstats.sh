#!/bin/bash
# Per-kernel durations of the bench's steady loop -> gpurun_out/kstats_<tag>.csv (run via gpurun):  tools/kstats.sh TAG [ENV=VAL ...]
# rocprofv3 --kernel-trace --stats around `bench.py --steps 96 --warmup 12 --repeats 1 --no-post --no-other --cpu-pairs 0`
tag=$1; shift
out=$GRAFT_REPO_ROOT/gpurun_out/kstats_$tag
mkdir -p $out
export TMPDIR=/tmp GPU_MAX_HW_QUEUES=24
for kv in "$@"; do export "$kv"; done
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out -o bench -- python3 $GRAFT_REPO_ROOT/bench.py --cpu-pairs 0 --no-post --no-other --repeats 1 --steps 96 --warmup 12 > $out/bench.json 2> $out/err.txt || { tail -5 $out/err.txt; exit 1; }
f=$(find $out -name "*kernel_stats.csv" | head -1)
cp $f $GRAFT_REPO_ROOT/gpurun_out/kstats_$tag.csv
python3 - <<PY
import csv,json
rows=list(csv.DictReader(open("$f")))
print("value", json.load(open("$out/bench.json"))["value"])
for r in rows[:26]:
    print("%-58s n=%5s avg_us=%8.1f pct=%s"%(r["Name"][:58], r["Calls"], float(r["AverageNs"])/1e3, r["Percentage"]))
PY
find $out -name "*.csv" ! -name "*kernel_stats.csv" -delete
