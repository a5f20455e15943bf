#!/bin/bash
# A/B of environment settings on the SAME box, back to back: tools/ab_env.sh STEPS "ENV=VAL ..." "ENV=VAL ..." ...
steps=$1; shift
for cfg in "$@"; do
  v=$(env $cfg python3 bench.py --steps $steps --warmup 12 --cpu-pairs 0 --no-post 2>/dev/null | python3 -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print(d['value'], d['ms_per_step'], d['accepted_frames'])")
  echo "[$cfg] steps=$steps -> $v"
done
