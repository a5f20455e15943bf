#!/bin/bash
# usage: tools/rep_bench.sh <reps> <steps> <warmup> [ENV=VAL ...] -- prints the values of <reps> bench runs
reps=$1; k=$2; w=$3; shift 3
mkdir -p gpurun_out
vals=""
for i in $(seq $reps); do
  env "$@" timeout -k 10 200 python bench.py --steps $k --warmup $w --cpu-pairs 0 --no-post > gpurun_out/rep.json 2> gpurun_out/rep.err || { echo "failed"; exit 1; }
  vals="$vals $(python -c "import json;print('%.0f' % json.load(open('gpurun_out/rep.json'))['value'])")"
done
echo "$* steps=$k:$vals"
