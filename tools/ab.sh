#!/bin/bash
# usage: ab.sh tag "ENV=.. " ...   runs bench quickly for each env setting, alternating, 2 rounds
out=gpurun_out/ab_$1; mkdir -p $out; shift
for round in 1 2; do
  i=0
  for e in "$@"; do
    i=$((i+1))
    env $e python bench.py --gpus 1 --steps 20 --warmup 5 --cpu-pairs 0 --no-other > $out/r${round}_v${i}.json 2>$out/r${round}_v${i}.err
    python - <<PY
import json
d=json.loads(open("$out/r${round}_v${i}.json").read().strip().splitlines()[-1])
print("round $round", "$e", d["value"], d["window_values"], d.get("steady_state",{}).get("value"), d.get("default_odometer",{}).get("value"), d.get("stage_ms_per_pair_alone"))
PY
  done
done
