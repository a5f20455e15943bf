#!/bin/bash
# Alternating A/B of environment settings on ONE box (via gpurun):  tools/ab.sh ROUNDS "ENV=VAL ..." "ENV=VAL ..." ...
# Every round runs the bench once per setting (windows, steady state, default odometer, from-host leg; no CPU legs, no other
# configs); at the end one line per setting: mean +- standard error of the 20-pair window value and of the steady rate, the
# means of the other two.  Boxes differ by +-3 %, single runs by +-2 %: this is the only comparison that resolves 1 %.
R=$1; shift
out=gpurun_out/ab_$$.txt
for r in $(seq 1 $R); do
for e in "$@"; do
  v=$(env $e python bench.py --gpus 1 --steps 20 --warmup 5 --cpu-pairs 0 --no-other 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['steady_state']['value'], d['default_odometer']['value'], d.get('from_host_pairs_per_s'))")
  echo "cfg [$e] $v"
done; done > $out 2>&1
python - <<PY
import collections, statistics, re
d=collections.defaultdict(list)
for l in open("$out"):
    m=re.match(r"cfg \[(.*)\] (.*)", l)
    if m: d[m.group(1)].append([float(x) for x in m.group(2).split()])
for k,v in d.items():
    c=list(zip(*v)); se=lambda x: statistics.pstdev(x)/len(x)**0.5
    print("%-32s n %d | window %.1f+-%.1f | steady %.1f+-%.1f | default-odo %.1f | from-host %.1f" % (k, len(v), statistics.mean(c[0]), se(c[0]), statistics.mean(c[1]), se(c[1]), statistics.mean(c[2]), statistics.mean(c[3])))
PY
