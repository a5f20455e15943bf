#!/bin/bash
# Per-kernel durations of the bench loop -> gpurun_out/kstats_<tag>{,_alone}.csv (run via gpurun):  tools/kstats.sh TAG [ENV=VAL ...]
#   mix:   rocprofv3 --kernel-trace --stats around `bench.py --steps 96 --warmup 12 --repeats 1 --no-post --no-other --cpu-pairs 0`
#          (16 pairs in flight: a kernel's duration here is its latency beside the other pairs' kernels)
#   alone: the same with VO_LOOKAHEAD=0 VO_POSE_AHEAD=0 (one pair at a time: every kernel alone on the GPU), 24 steps
tag=$1; shift
out=$GRAFT_REPO_ROOT/gpurun_out/kstats_$tag
mkdir -p $out
export TMPDIR=/tmp GPU_MAX_HW_QUEUES=24
for kv in "$@"; do export "$kv"; done
cd /tmp
B="python3 $GRAFT_REPO_ROOT/bench.py --cpu-pairs 0 --no-post --no-other --repeats 1"
rocprofv3 --kernel-trace --stats --output-format csv -d $out/mix -o bench -- $B --steps 96 --warmup 12 > $out/bench.json 2> $out/err.txt || { tail -5 $out/err.txt; exit 1; }
VO_LOOKAHEAD=0 VO_POSE_AHEAD=0 rocprofv3 --kernel-trace --stats --output-format csv -d $out/alone -o bench -- $B --steps 24 --warmup 4 > $out/bench_alone.json 2> $out/err_alone.txt || { tail -5 $out/err_alone.txt; exit 1; }
cp $(find $out/mix -name "*kernel_stats.csv" | head -1) $GRAFT_REPO_ROOT/gpurun_out/kstats_$tag.csv
cp $(find $out/alone -name "*kernel_stats.csv" | head -1) $GRAFT_REPO_ROOT/gpurun_out/kstats_${tag}_alone.csv
python3 - <<PY
import csv,json
mix={r["Name"]:r for r in csv.DictReader(open("$GRAFT_REPO_ROOT/gpurun_out/kstats_$tag.csv"))}
alone={r["Name"]:r for r in csv.DictReader(open("$GRAFT_REPO_ROOT/gpurun_out/kstats_${tag}_alone.csv"))}
print("value (mix run)", json.load(open("$out/bench.json"))["value"], " value (alone run)", json.load(open("$out/bench_alone.json"))["value"])
print("%-58s %6s %10s %10s"%("kernel","calls","mix_us","alone_us"))
for n,r in list(mix.items())[:28]:
    a=alone.get(n)
    print("%-58s %6s %10.1f %10s"%(n[:58], r["Calls"], float(r["AverageNs"])/1e3, "%.1f"%(float(a["AverageNs"])/1e3) if a else "-"))
PY
rm -rf $out/mix $out/alone
