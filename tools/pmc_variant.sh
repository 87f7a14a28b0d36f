#!/bin/bash
# Development aid: VALU / LDS / SALU instructions per wave and duration of the stack kernel for a variant library.
#   tools/pmc_variant.sh <label> [APGPU_LIBRARY path] [bench.py args...]      (run through gpurun)
LABEL=$1; LIB=$2; shift 2
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/pmcv_$LABEL
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
[ -n "$LIB" ] && export APGPU_LIBRARY=$REPO/$LIB
rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES SQ_INSTS_LDS SQ_INSTS_SALU --output-format csv -d $OUT/p1 -- python3 $REPO/bench.py --steps 3 --warmup 1 --no-cpu-baseline "$@" > $OUT/b1.json 2> $OUT/p1.log
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/t -- python3 $REPO/bench.py --steps 10 --warmup 2 --no-cpu-baseline "$@" > $OUT/b2.json 2> $OUT/t.log
python3 - <<PY
import csv, glob
c = {}
for f in glob.glob('$OUT/p1/*/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        if 'stack_' in r['Kernel_Name']:
            c.setdefault(r['Counter_Name'], []).append(float(r['Counter_Value']))
a = {k: sum(v) / len(v) for k, v in c.items()}
d = []
for f in glob.glob('$OUT/t/*/*kernel_trace.csv'):
    for r in csv.DictReader(open(f)):
        if 'stack_' in r['Kernel_Name']:
            d.append(int(r['End_Timestamp']) - int(r['Start_Timestamp']))
w = a.get('SQ_WAVES', 1)
print('$LABEL: VALU/wave %.0f  SALU/wave %.0f  LDS/wave %.0f  waves %d  avg %.4f ms  min %.4f ms (%d launches)' % (
    a.get('SQ_INSTS_VALU', 0) / w, a.get('SQ_INSTS_SALU', 0) / w, a.get('SQ_INSTS_LDS', 0) / w, w, sum(d) / len(d) / 1e6, min(d) / 1e6, len(d)))
PY
