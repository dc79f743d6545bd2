#!/bin/bash
# A/B builds: montecarlooptionspricer_amd/lib/libmcgpu_<name>.so = the in-tree objects with ONE translation unit
# recompiled under extra flags (tools/ab_libs.sh alternates between such libraries on one box).
#   tools/build_variant.sh gt2 kernels_gbm.hip -DMCG_GBM_TABLES=2
set -e
NAME=$1; TU=$2; shift 2
PKG=montecarlooptionspricer_amd
make -s lib
mkdir -p build/obj_$NAME
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Iinclude -Wall -Wno-unused-function "$@" \
    -c $PKG/csrc/$TU -o build/obj_$NAME/$TU.o
OBJS=$(ls build/obj/*.o | grep -v "/$TU.o")
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o $PKG/lib/libmcgpu_$NAME.so $OBJS build/obj_$NAME/$TU.o -ldl
echo "built $PKG/lib/libmcgpu_$NAME.so"
