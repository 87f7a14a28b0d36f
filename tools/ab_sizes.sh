#!/bin/bash
# usage: ab_sizes.sh "<sizes>" variant...
SIZES=$1; shift
REPO=${GRAFT_REPO_ROOT:-/root/repo}
for n in $SIZES; do
  for v in "$@"; do
    if [ "$v" = prod ]; then unset APGPU_LIBRARY; else export APGPU_LIBRARY=$REPO/build_variants/$v/libapgpu.so; fi
    python3 $REPO/bench.py --no-cpu-baseline --steps 20 --frames $n 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('N=$n $v', round(d['roofline']['avg_launch_ms'],4), round(d['roofline']['min_launch_ms'],4))"
  done
done
