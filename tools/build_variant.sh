#!/bin/bash
# build a variant of the library with extra flags for ONE source file:
#   tools/build_variant.sh NAME FILE "-DFLAG=1"      e.g.  tools/build_variant.sh c512 mapper "-DGS_MAP_CHUNK=512"
set -e
cd "$(dirname "$0")/../taichi_gaussian_rasterizer_amd/csrc"
mkdir -p ../../tools/ubench/bin
EXTRA=""
case $2 in mapper) EXTRA="-ffp-contract=off";; raster_fwd|raster_bwd) EXTRA="-fno-slp-vectorize";; esac
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -fno-fast-math $EXTRA $3 -c $2.hip -o _obj/$2_$1.o
OBJS=$(ls _obj/*.o | grep -v "_obj/$2" | tr '\n' ' ')
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/ubench/bin/lib_$1.so $OBJS _obj/$2_$1.o
rm _obj/$2_$1.o
echo built tools/ubench/bin/lib_$1.so
