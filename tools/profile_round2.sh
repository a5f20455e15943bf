#!/bin/bash
# Round-2 profile set on the GPU box -> gpurun_out/prof_<tag>/ :  usage: tools/profile_round2.sh TAG   (run via gpurun)
#   stats/        rocprofv3 --kernel-trace --stats of the default bench command
#   fetch_*/ write_*/   the two HBM-traffic PMC passes (separate runs), line scheme and raster scheme (VO_RASTER=1)
#   occ_*/        SQ occupancy / issue / stall counters of the same runs
set -e
tag=${1:-r02}
out=$GRAFT_REPO_ROOT/gpurun_out/prof_$tag
mkdir -p $out
export TMPDIR=/tmp
export GPU_MAX_HW_QUEUES=24   # in the shell: under rocprofv3 the runtime starts before python can set it
cd /tmp
B="python3 $GRAFT_REPO_ROOT/bench.py --cpu-pairs 0 --no-post"
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -o bench -- $B --steps 96 --warmup 12 > $out/bench_under_rocprof.json 2> $out/stats.err
echo "stats done"
export VO_WE_FUSE=0          # schemes 0 / 1: separate W and E volumes (0 = line scheme, 1 = raster scheme); scheme 2: line scheme with W+E fused
for scheme in 0 1 2; do
  if [ $scheme = 2 ]; then export VO_RASTER=0; export VO_WE_FUSE=1; else export VO_RASTER=$scheme; fi
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $out/fetch_$scheme -o pmc -- $B --steps 6 --warmup 2 > /dev/null 2> $out/fetch_$scheme.err
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $out/write_$scheme -o pmc -- $B --steps 6 --warmup 2 > /dev/null 2> $out/write_$scheme.err
  rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $out/occ_$scheme -o pmc -- $B --steps 6 --warmup 2 > /dev/null 2> $out/occ_$scheme.err
  echo "pmc scheme $scheme done"
done
find $out -name "*.csv" | head -40
