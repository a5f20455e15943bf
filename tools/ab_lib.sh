#!/bin/bash
# usage: tools/ab_lib.sh <tag> <lib.so> [ENV=VAL ...] -- 20/5 and 240/12 bench against the given library build
tag=$1; lib=$2; shift 2
mkdir -p gpurun_out
for spec in "20 5" "240 12"; do
  set -- $spec "$@"
  k=$1; w=$2; shift 2
  env "$@" timeout -k 10 200 python tools/bench_with_lib.py $lib --steps $k --warmup $w --cpu-pairs 0 --no-post > gpurun_out/ab_${tag}_$k.json 2> gpurun_out/ab_${tag}_$k.err || exit 1
  python - <<PY
import json
d=json.load(open("gpurun_out/ab_${tag}_$k.json"))
print("${tag} steps=$k value=%.1f" % d["value"])
PY
done
