#!/bin/bash
# The floor table of DESIGN 4.1 on the kernel that ships: for the product library and the stripped fast
# kernels (tools/variant_lib.sh strip1 / strip2), the fast kernel's duration under rocprofv3 and its VALU counters.
#   bash tools/pmc_floor.sh <label> [APGPU_LIBRARY path relative to the repo]      (run through gpurun)
LABEL=$1; LIB=$2
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/pmcf_$LABEL
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
[ -n "$LIB" ] && export APGPU_LIBRARY=$REPO/$LIB
rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES SQ_ACTIVE_INST_VALU SQ_INSTS_SALU --output-format csv -d $OUT/p1 -- python3 $REPO/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/b1.json 2> $OUT/p1.log
rocprofv3 --pmc GRBM_GUI_ACTIVE --output-format csv -d $OUT/p2 -- python3 $REPO/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/b2.json 2> $OUT/p2.log
rocprofv3 --kernel-trace --output-format csv -d $OUT/t -- python3 $REPO/bench.py --steps 10 --warmup 2 --no-cpu-baseline > $OUT/b3.json 2> $OUT/t.log
python3 - <<PY
import csv, glob
c = {}
for f in glob.glob('$OUT/p[12]/*/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        if 'stack_fast_kernel' in r['Kernel_Name']:
            c.setdefault(r['Counter_Name'], []).append(float(r['Counter_Value']))
a = {k: sum(v) / len(v) for k, v in c.items()}
d = []
for f in glob.glob('$OUT/t/*/*kernel_trace.csv'):
    for r in csv.DictReader(open(f)):
        if 'stack_fast_kernel' in r['Kernel_Name']:
            d.append(int(r['End_Timestamp']) - int(r['Start_Timestamp']))
w = a.get('SQ_WAVES', 1)
avg = sum(d) / len(d)
busy = a.get('SQ_ACTIVE_INST_VALU', 0) * 4.0 / (1024.0 * a.get('GRBM_GUI_ACTIVE', 1) / 8.0)
print('%-8s VALU/wave %5.0f  SALU/wave %4.0f  VALU busy %.3f  clock %.3f GHz  fast kernel avg %.4f ms  min %.4f ms (%d launches)  %.1f %% of 8 TB/s at the average' % (
    '$LABEL', a.get('SQ_INSTS_VALU', 0) / w, a.get('SQ_INSTS_SALU', 0) / w, busy, a.get('GRBM_GUI_ACTIVE', 0) / 8.0 / avg, avg / 1e6, min(d) / 1e6, len(d),
    100 * 4563402752 / (avg * 1e-9) / 8e12))
PY
rm -rf $OUT/p1 $OUT/p2 $OUT/t
