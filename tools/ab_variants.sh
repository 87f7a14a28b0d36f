#!/bin/bash
# Development aid: interleaved A/B timing of variant libraries on one box (run through gpurun).
#   tools/ab_variants.sh <rounds> <variant> [<variant> ...]     ("prod" = the in-tree library)
# Each round runs bench.py once per variant (30 timed steps); prints per-variant median / min of the per-process
# average and minimum kernel times, so that box-to-box and run-to-run drift cancels.
ROUNDS=$1; shift
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/ab; mkdir -p $OUT; rm -f $OUT/*.txt
for r in $(seq $ROUNDS); do
  for v in "$@"; do
    if [ "$v" = prod ]; then unset APGPU_LIBRARY; else export APGPU_LIBRARY=$REPO/build_variants/$v/libapgpu.so; fi
    python3 $REPO/bench.py --no-cpu-baseline --steps 30 $AB_ARGS 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['roofline']['avg_launch_ms'], d['roofline']['min_launch_ms'])" >> $OUT/$v.txt
  done
done
python3 - "$@" <<PY
import sys
for v in sys.argv[1:]:
    rows = [tuple(map(float, l.split())) for l in open('$OUT/%s.txt' % v)]
    avg = sorted(r[0] for r in rows); mn = sorted(r[1] for r in rows)
    print('%-10s avg: median %.4f min %.4f   min-launch: median %.4f min %.4f   (%d runs)' % (v, avg[len(avg)//2], avg[0], mn[len(mn)//2], mn[0], len(rows)))
PY
