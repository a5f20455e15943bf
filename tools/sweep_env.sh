#!/bin/bash
# usage: tools/sweep_env.sh "VAR1=a VAR2=b" "VAR1=c" ...   -> bench value per environment (same box, back to back)
for cfg in "$@"; do
  v=$(env $cfg python bench.py --steps ${SWEEP_STEPS:-300} --warmup 20 --cpu-pairs 0 --no-events 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['value'])")
  echo "$cfg -> $v"
done
