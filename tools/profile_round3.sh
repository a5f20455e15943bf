#!/bin/bash
# Round-3 profile set on the GPU box -> gpurun_out/prof_<tag>/ :  usage: tools/profile_round3.sh TAG [quick|full] [WORKLOAD]   (run via gpurun)
#   stats/            rocprofv3 --kernel-trace --stats of the default bench command (driver window: --steps 20 --warmup 5, and 96 steps)
#   fetch/ write/     the two HBM-traffic PMC passes (separate runs) of the default schedule
#   occ/              SQ occupancy / issue / stall counters of the same run
set -e
tag=${1:-r03}
out=$GRAFT_REPO_ROOT/gpurun_out/prof_$tag
mkdir -p $out
export TMPDIR=/tmp
export GPU_MAX_HW_QUEUES=24   # in the shell: under rocprofv3 the runtime starts before python can set it
cd /tmp
WL=${3:-C2}
B="python3 $GRAFT_REPO_ROOT/bench.py --cpu-pairs 0 --no-post --no-other --workload $WL"
N1=96; W1=12
if [ "$WL" != "C2" ]; then N1=24; W1=4; fi
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -o bench -- $B --steps $N1 --warmup $W1 > $out/bench_under_rocprof.json 2> $out/stats.err
echo "stats done"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $out/fetch -o pmc -- $B --steps 6 --warmup 2 > /dev/null 2> $out/fetch.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $out/write -o pmc -- $B --steps 6 --warmup 2 > /dev/null 2> $out/write.err
echo "traffic done"
if [ "$2" != "quick" ]; then
  rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $out/occ -o pmc -- $B --steps 6 --warmup 2 > /dev/null 2> $out/occ.err
  echo "occupancy done"
fi
find $out -name "*.csv" | head -40
