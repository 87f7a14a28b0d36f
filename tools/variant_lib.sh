#!/bin/bash
# Development aid: builds a VARIANT of libapgpu.so for same-box A/B measurements.
#   tools/variant_lib.sh <name> "<extra hipcc flags>" <translation unit> [<translation unit> ...]
# recompiles the named csrc/*.hip files with the extra flags (e.g. -DAPGPU_VARIANT_X) into build_variants/<name>/ and links
# them with the production objects of every other unit -> build_variants/<name>/libapgpu.so.  Select it at run time with
# APGPU_LIBRARY=build_variants/<name>/libapgpu.so (astrophotography_amd/_lib.py).  Never used by tests or the bench.
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
NAME=$1; FLAGS=$2; shift 2
OUT=$ROOT/build_variants/$NAME
mkdir -p $OUT
CS=$ROOT/astrophotography_amd/csrc
OBJS=""
for o in $CS/_obj/*.o; do
  b=$(basename $o .o); skip=0
  for tu in "$@"; do [ "$b" = "$(basename $tu .hip)" ] && skip=1; done
  [ $skip = 0 ] && OBJS="$OBJS $o"
done
for tu in "$@"; do
  b=$(basename $tu .hip)
  EXTRA=""; case $b in stack_inst_*) EXTRA="-mllvm -disable-machine-licm";; esac     # as _build.py's STACK_TU_FLAGS
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wall -Wno-unused-function \
    -I$ROOT/include -I$CS $EXTRA $FLAGS -c $CS/$b.hip -o $OUT/$b.o &
done
wait
for tu in "$@"; do OBJS="$OBJS $OUT/$(basename $tu .hip).o"; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT/libapgpu.so $OBJS
echo built $OUT/libapgpu.so
