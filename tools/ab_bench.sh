#!/bin/bash
# usage: tools/ab_bench.sh <tag> [ENV=VAL ...] -- runs the 20/5 and 240/12 bench with the given environment, prints value
tag=$1; shift
mkdir -p gpurun_out
for spec in "20 5" "240 12"; do
  set -- $spec "$@"
  k=$1; w=$2; shift 2
  env "$@" timeout -k 10 200 python bench.py --steps $k --warmup $w --cpu-pairs 0 --no-post > gpurun_out/ab_${tag}_$k.json 2> gpurun_out/ab_${tag}_$k.err || exit 1
  python - <<PY
import json
d=json.load(open("gpurun_out/ab_${tag}_$k.json"))
print("${tag} steps=$k value=%.1f alone_us=%s" % (d["value"], d["roofline"].get("alone", {})))
PY
done
