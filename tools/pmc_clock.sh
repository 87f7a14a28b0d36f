#!/bin/bash
# Development aid: the clock the chip holds during the stack kernel: GRBM_GUI_ACTIVE (summed over 8 XCDs) / 8 / duration.
#   tools/pmc_clock.sh <label> [APGPU_LIBRARY path] [bench.py args...]      (run through gpurun)
LABEL=$1; LIB=$2; shift 2
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/clk_$LABEL
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
[ -n "$LIB" ] && export APGPU_LIBRARY=$REPO/$LIB
rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/p -- python3 $REPO/bench.py --steps 6 --warmup 2 --no-cpu-baseline "$@" > $OUT/b.json 2> $OUT/p.log
python3 - <<PY
import csv, glob
c = []; d = {}
for f in glob.glob('$OUT/p/*/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        if 'stack_' in r['Kernel_Name'] and r['Counter_Name'] == 'GRBM_GUI_ACTIVE':
            c.append((r.get('Dispatch_Id'), float(r['Counter_Value'])))
for f in glob.glob('$OUT/p/*/*kernel_trace.csv'):
    for r in csv.DictReader(open(f)):
        if 'stack_' in r['Kernel_Name']:
            d[r.get('Dispatch_Id')] = int(r['End_Timestamp']) - int(r['Start_Timestamp'])
vals = [(v / 8.0 / d[k]) for k, v in c if k in d]
durs = [d[k] for k, _ in c if k in d]
print('$LABEL: clock %.3f GHz (min %.3f max %.3f), duration under counters avg %.4f ms, %d dispatches' % (
    sum(vals) / len(vals), min(vals), max(vals), sum(durs) / len(durs) / 1e6, len(vals)))
PY
