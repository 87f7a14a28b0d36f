#!/bin/bash
# Development aid: links a variant of libapgpu.so in which ONE translation unit is rebuilt with extra -D flags.
#   bash tools/variant_lib.sh <tag> <source.hip> [-DFLAG ...]   -> build_variants/libapgpu_<tag>.so
# Use it with  APGPU_LIBRARY=<path> python tools/bench_kernels.py ...  (the loader honours APGPU_LIBRARY).
set -e
TAG=$1; SRC=$2; shift; shift
REPO=$(cd "$(dirname "$0")/.." && pwd)
CS=$REPO/astrophotography_amd/csrc
OUT=$REPO/build_variants
mkdir -p $OUT
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -I$CS -I$REPO/include "$@" -c $CS/$SRC -o $OUT/${SRC%.hip}_$TAG.o
OBJS=""
for o in $CS/_obj/*.o; do
  if [ "$(basename $o)" != "${SRC%.hip}.o" ]; then OBJS="$OBJS $o"; fi
done
hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT/libapgpu_$TAG.so $OBJS $OUT/${SRC%.hip}_$TAG.o
echo $OUT/libapgpu_$TAG.so
