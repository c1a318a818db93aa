#!/bin/bash
# Dev tool: build a tuning variant of libgnxr.so -- api.hip recompiled with extra flags, the other objects reused from build/.
# usage: tools/build_variant.sh <name> "<extra hipcc flags>"   ->  ab_libs/lib_<name>.so   (select it with GNXR_LIB=..., tests/dev_ab.py)
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
NAME=$1; shift
mkdir -p $ROOT/ab_libs $ROOT/build
cd $ROOT/gnxraytracer_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-slp-vectorize -fno-unroll-loops -Wall -Wno-unused-variable -Wno-unused-function $@ -c api.hip -o $ROOT/build/api_$NAME.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $ROOT/build/inst_whitted_tex.o $ROOT/build/inst_whitted.o $ROOT/build/inst_vol.o $ROOT/build/api_$NAME.o $ROOT/build/scene_compile.o $ROOT/build/scene_builder.o -o $ROOT/ab_libs/lib_$NAME.so -lpthread
echo built $ROOT/ab_libs/lib_$NAME.so
