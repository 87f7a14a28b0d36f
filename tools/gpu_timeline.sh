#!/bin/bash
# development aid: start / end timestamps of every dispatch of a few bench steps (through gpurun)
R=$PWD
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d /tmp/tl -- python3 $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline "$@" > /dev/null 2>&1
f=$(find /tmp/tl -name "*kernel_trace.csv" | head -1)
python3 - $f <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'stack_fast' in r['Kernel_Name'] or 'stack_redo' in r['Kernel_Name']]
lo = max(idx[-12] - 2, 0) if len(idx) >= 12 else 0
t0 = int(rows[lo]['Start_Timestamp'])
for r in rows[lo:]:
    s, e = int(r['Start_Timestamp']) - t0, int(r['End_Timestamp']) - t0
    print('%10.1f us  +%8.1f us  %s' % (s / 1e3, (e - s) / 1e3, r['Kernel_Name'][:90]))
PY
