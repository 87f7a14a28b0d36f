#!/bin/bash
O=$PWD/gpurun_out/r05r; mkdir -p $O; rm -f $O/*.txt
R=$PWD
for v in prod quiet8 prod quiet8; do
  if [ $v = prod ]; then unset APGPU_LIBRARY; else export APGPU_LIBRARY=$R/build_variants/$v/libapgpu.so; fi
  python bench.py --workload c5 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', d['ms_per_step'], d['redo'])" >> $O/c5.txt
done
cat $O/c5.txt
