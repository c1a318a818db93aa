#!/bin/bash
# Dev tool: build a tuning variant of libgnxr.so -- api.hip recompiled with extra flags, the other objects reused from build/
# (ALLTU=1: the explicit-instantiation units inst_*.hip are recompiled with the flags too).
# usage: [ALLTU=1] tools/build_variant.sh <name> "<extra hipcc flags>"   ->  ab_libs/lib_<name>.so   (select it with GNXR_LIB=..., tests/dev_ab.py)
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
NAME=$1; shift
mkdir -p $ROOT/ab_libs $ROOT/build
cd $ROOT/gnxraytracer_amd/csrc
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-slp-vectorize -fno-unroll-loops -Wall -Wno-unused-variable -Wno-unused-function"
OBJS=""
if [ -n "$ALLTU" ]; then
  for f in inst_whitted_tex inst_whitted inst_vol; do /opt/rocm/bin/hipcc $FLAGS $@ -c $f.hip -o $ROOT/build/${f}_$NAME.o & OBJS="$OBJS $ROOT/build/${f}_$NAME.o"; done
else
  OBJS="$ROOT/build/inst_whitted_tex.o $ROOT/build/inst_whitted.o $ROOT/build/inst_vol.o"
fi
/opt/rocm/bin/hipcc $FLAGS $@ -c api.hip -o $ROOT/build/api_$NAME.o
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OBJS $ROOT/build/api_$NAME.o $ROOT/build/scene_compile.o $ROOT/build/scene_builder.o -o $ROOT/ab_libs/lib_$NAME.so -lpthread
echo built $ROOT/ab_libs/lib_$NAME.so
