#!/bin/bash
# build a variant of the library with extra mapper flags:  tools/build_variant.sh NAME "-DGS_MAP_CHUNK=512"
set -e
cd "$(dirname "$0")/../taichi_gaussian_rasterizer_amd/csrc"
mkdir -p ../../tools/ubench/bin
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -fno-fast-math -ffp-contract=off $2 -c mapper.hip -o _obj/mapper_$1.o
OBJS=$(ls _obj/*.o | grep -v "mapper" | tr '\n' ' ')
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/ubench/bin/lib_$1.so $OBJS _obj/mapper_$1.o
rm _obj/mapper_$1.o
echo built tools/ubench/bin/lib_$1.so
