#!/bin/bash
# usage: tools/sweep_env.sh "<steps> <warmup>" VAR "v1 v2 ..." [ENV=VAL ...]  -- bench value per setting of VAR
spec=$1; var=$2; vals=$3; shift 3
set -- $spec "$@"; k=$1; w=$2; shift 2
mkdir -p gpurun_out
for v in $vals; do
  env "$@" $var=$v timeout -k 10 200 python bench.py --steps $k --warmup $w --cpu-pairs 0 --no-post > gpurun_out/sw.json 2> gpurun_out/sw.err || { echo "$var=$v failed"; tail -2 gpurun_out/sw.err; continue; }
  python -c "import json;d=json.load(open('gpurun_out/sw.json'));print('$* $var=$v steps=$k value=%.1f' % d['value'])"
done
