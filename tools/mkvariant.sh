#!/bin/bash
# A/B build of the library with extra -D flags on sgbm.hip: tools/mkvariant.sh <tag> [-DNAME=VALUE ...]  ->  build/libvo355_<tag>.so
# (run it against the same box with VO355_LIB=build/libvo355_<tag>.so)
set -e
tag=$1; shift
cd "$(dirname "$0")/../openvo_amd/csrc"
mkdir -p ../../build
make -s
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wno-unused-function -Wno-unused-variable "$@" -c sgbm.hip -o ../../build/sgbm_$tag.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../build/libvo355_$tag.so vo_ctx.o ../../build/sgbm_$tag.o orb.o match.o geom.o ransac.o mgpu.o -ldl
ls -la ../../build/libvo355_$tag.so
