#!/bin/bash
O=$PWD/gpurun_out/r05q; mkdir -p $O; rm -f $O/*.txt
R=$PWD
cd /tmp && export TMPDIR=/tmp
for v in prod relaxed prod relaxed; do
  if [ $v = prod ]; then unset APGPU_LIBRARY; else export APGPU_LIBRARY=$R/build_variants/$v/libapgpu.so; fi
  rm -rf /tmp/tr_$v
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/tr_$v -- python3 $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline > $O/line_$v.json 2>/dev/null
  f=$(find /tmp/tr_$v -name "*kernel_stats.csv" | head -1)
  echo "== $v: $(python3 -c "import json;d=json.load(open('$O/line_$v.json'));print(d['ms_per_step'])") ms per step" >> $O/redo_pass.txt
  python3 -c "
import csv,sys
for r in csv.DictReader(open('$f')):
    if 'stack_' in r['Name']: print('   %-90s calls %s avg %.1f us min %.1f max %.1f' % (r['Name'].replace('apgpu_stack::','')[:90], r['Calls'], float(r['AverageNs'])/1e3, float(r['MinNs'])/1e3, float(r['MaxNs'])/1e3))
" >> $O/redo_pass.txt
done
unset APGPU_LIBRARY
cd $R
cat $O/redo_pass.txt
APGPU_LIBRARY=$R/build_variants/relaxed/libapgpu.so python -m pytest tests/test_gpu_redo.py -m gpu -x -q 2>&1 | tail -2
