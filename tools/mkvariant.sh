#!/bin/bash
# A/B build of the library with extra -D flags on one source file (default sgbm.hip):
#   tools/mkvariant.sh <tag> [-s orb.hip] [-DNAME=VALUE ...]  ->  build/libvo355_<tag>.so
# (run it against the same box with VO355_LIB=build/libvo355_<tag>.so)
set -e
tag=$1; shift
src=sgbm.hip
if [ "$1" = "-s" ]; then src=$2; shift 2; fi
base=${src%.hip}
cd "$(dirname "$0")/../openvo_amd/csrc"
mkdir -p ../../build
make -s
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wno-unused-function -Wno-unused-variable "$@" -c $src -o ../../build/${base}_$tag.o
objs=""
for o in vo_ctx sgbm orb match geom ransac mgpu; do
  if [ "$o" = "$base" ]; then objs="$objs ../../build/${base}_$tag.o"; else objs="$objs $o.o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../build/libvo355_$tag.so $objs -ldl
ls -la ../../build/libvo355_$tag.so
