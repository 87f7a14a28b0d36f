#!/bin/bash
O=gpurun_out/r05f; mkdir -p $O
R=$PWD
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/trace -- python3 $R/tools/redo_sweep.py --rounds 1 --steps 6 --cases "100 %" > $R/$O/sweep_traced.txt 2> $R/$O/trace.log
cd $R
f=$(find $O/trace -name "*kernel_stats.csv" | head -1); grep "stack_" $f | cut -c1-220
t=$(find $O/trace -name "*kernel_trace.csv" | head -1)
python3 - $t <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
rows=[r for r in rows if 'stack' in r['Kernel_Name']]
rows.sort(key=lambda r:int(r['Start_Timestamp']))
prev=None
for r in rows[:14]+rows[-8:]:
    s,e=int(r['Start_Timestamp']),int(r['End_Timestamp'])
    print('%-70s grid %8s dur %8.1f us  gap %8.1f us'%(r['Kernel_Name'][5:75],r['Grid_Size'],(e-s)/1e3,(s-prev)/1e3 if prev else 0))
    prev=e
PY
rm -rf $O/trace
AB_ARGS="" bash tools/ab_variants.sh 4 prod strip1 strip2 2>&1 | tee $O/ab_bench.txt
