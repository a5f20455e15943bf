#!/bin/bash
# Collect the judged profile set on the GPU box into gpurun_out/prof_<tag>/ : kernel stats of the bench
# command, and the two PMC passes.  usage: tools/profile_round.sh TAG     (run via gpurun)
set -e
tag=${1:-r01}
out=$GRAFT_REPO_ROOT/gpurun_out/prof_$tag
mkdir -p $out
export TMPDIR=/tmp
export GPU_MAX_HW_QUEUES=24   # in the shell: under rocprofv3 the runtime starts before python can set it
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -o bench -- python3 $GRAFT_REPO_ROOT/bench.py --steps 48 --warmup 6 --cpu-pairs 0 > $out/bench_under_rocprof.json 2> $out/stats.err
echo "stats done"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $out/fetch -o pmc -- python3 $GRAFT_REPO_ROOT/bench.py --steps 8 --warmup 2 --cpu-pairs 0 > /dev/null 2> $out/fetch.err
echo "fetch done"
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $out/write -o pmc -- python3 $GRAFT_REPO_ROOT/bench.py --steps 8 --warmup 2 --cpu-pairs 0 > /dev/null 2> $out/write.err
echo "write done"
ls $out/*
