#!/bin/bash
O=$PWD/gpurun_out/r05s; mkdir -p $O; rm -f $O/*.txt
R=$PWD
for v in prod acqrel prod acqrel; do
  if [ $v = prod ]; then unset APGPU_LIBRARY; else export APGPU_LIBRARY=$R/build_variants/$v/libapgpu.so; fi
  echo "== $v" >> $O/sweep.txt
  python tools/redo_sweep.py --cases "natural,10 %,25 %,100" 2>/dev/null | grep -v "^#" | cut -c1-150 >> $O/sweep.txt
done
cat $O/sweep.txt
