#!/bin/bash
# The judged profile set of a round, on the GPU box -> gpurun_out/prof_<tag>/ ; reduce it with tools/pmc_summary.py <tag>.
#   usage (via gpurun): tools/profile_round.sh TAG [quick|full] [WORKLOAD]
#   stats/        rocprofv3 --kernel-trace --stats of `bench.py --steps 96 --warmup 12` (16 pairs in flight) -> kernel_stats
#   alone/        the same with VO_LOOKAHEAD=0 VO_POSE_AHEAD=0 (one pair at a time: every kernel alone on the GPU)
#   alone_engine/ the same with VO_LOOKAHEAD=1 VO_POSE_AHEAD=0: one pair at a time THROUGH A LOOK-AHEAD ENGINE, i.e. the kernel variants
#                 the timed region runs (the wide-strip diagonal sweep), each alone on the GPU -- the rocprof twin of the line's launch_us
#   fetch/ write/ the two HBM-traffic PMC passes (separate runs, --pmc with --kernel-trace only)
#   occ/          SQ occupancy / issue / stall counters (full only)
set -e
tag=${1:-r04}
out=$GRAFT_REPO_ROOT/gpurun_out/prof_$tag
mkdir -p $out
export TMPDIR=/tmp
export GPU_MAX_HW_QUEUES=24   # in the shell: under rocprofv3 the runtime starts before python can set it
# what the kernels were when these counters were taken: bench.py compares it with the sources it runs beside
python3 -c "import sys; sys.path.insert(0, '$GRAFT_REPO_ROOT'); from openvo_amd._native import csrc_digest; print(csrc_digest())" > $out/csrc_digest.txt
cd /tmp
WL=${3:-C2}
B="python3 $GRAFT_REPO_ROOT/bench.py --cpu-pairs 0 --no-post --no-other --repeats 1 --workload $WL"
N1=96; W1=12
if [ "$WL" != "C2" ]; then N1=24; W1=4; fi
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -o bench -- $B --steps $N1 --warmup $W1 > $out/bench_under_rocprof.json 2> $out/stats.err
VO_LOOKAHEAD=0 VO_POSE_AHEAD=0 rocprofv3 --kernel-trace --stats --output-format csv -d $out/alone -o bench -- $B --steps 24 --warmup 4 > $out/bench_alone_under_rocprof.json 2> $out/alone.err
VO_LOOKAHEAD=1 VO_POSE_AHEAD=0 rocprofv3 --kernel-trace --stats --output-format csv -d $out/alone_engine -o bench -- $B --steps 24 --warmup 4 > /dev/null 2> $out/alone_engine.err
echo "stats done"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $out/fetch -o pmc -- $B --steps 6 --warmup 2 > /dev/null 2> $out/fetch.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $out/write -o pmc -- $B --steps 6 --warmup 2 > /dev/null 2> $out/write.err
echo "traffic done"
if [ "$2" != "quick" ]; then
  rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $out/occ -o pmc -- $B --steps 6 --warmup 2 > /dev/null 2> $out/occ.err
  echo "occupancy done"
fi
# (the per-dispatch traces are large and not needed by the summary)
find $out -name "*kernel_trace.csv" -delete
find $out -name "*.csv" | head -40
