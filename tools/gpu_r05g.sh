#!/bin/bash
O=gpurun_out/r05g; mkdir -p $O
R=$PWD
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $R/$O/trace -- python3 $R/tools/redo_sweep.py --rounds 1 --steps 3 --cases "100 %" > $R/$O/sweep_traced.txt 2> $R/$O/trace.log
cd $R
t=$(find $O/trace -name "*kernel_trace.csv" | head -1)
head -1 $t
python3 - $t <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
rows=[r for r in rows if 'stack' in r['Kernel_Name']]
rows.sort(key=lambda r:int(r['Start_Timestamp']))
prev=None
for r in rows:
    s,e=int(r['Start_Timestamp']),int(r['End_Timestamp'])
    print('%-62s grid %s wg %s vgpr %s lds %s dur %8.1f us  gap %8.1f us'%(r['Kernel_Name'][5:67],r.get('Grid_Size_X',r.get('Grid_Size')),r.get('Workgroup_Size_X'),r.get('VGPR_Count'),r.get('LDS_Block_Size'),(e-s)/1e3,(s-prev)/1e3 if prev else 0))
    prev=e
PY
rm -rf $O/trace
