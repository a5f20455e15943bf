#!/bin/bash
# usage: tools/iso_kernels.sh TAG [ENV=VAL ...]  -> isolated (no look-ahead) per-kernel times of the bench loop
tag=$1; shift
for kv in "$@"; do export "$kv"; done
export VO_LOOKAHEAD=0 TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/iso_$tag -o iso -- python3 $GRAFT_REPO_ROOT/bench.py --steps 24 --warmup 4 --cpu-pairs 0 > /dev/null 2>&1
echo "== $tag $@"; head -${ISO_TOP:-4} $GRAFT_REPO_ROOT/gpurun_out/iso_$tag/iso_kernel_stats.csv | cut -d, -f1-4 | cut -c1-60,100-
