#!/bin/bash
O=gpurun_out/r05e; mkdir -p $O
R=$PWD
for v in bail8 nobail bailh; do
  export APGPU_LIBRARY=$R/build_variants/$v/libapgpu.so
  echo "== $v" >> $O/ab.txt
  python tools/redo_sweep.py --rounds 1 --steps 6 --cases 10,25,50,75,100 >> $O/ab.txt 2>> $O/ab.log
done
cat $O/ab.txt
export APGPU_LIBRARY=$R/build_variants/bail8/libapgpu.so
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/trace -- python3 $R/tools/redo_sweep.py --rounds 1 --steps 6 --cases "100 %" > $R/$O/sweep_traced.txt 2> $R/$O/trace.log
cd $R
f=$(find $O/trace -name "*kernel_stats.csv" | head -1); grep "stack_" $f | cut -c1-200
rm -rf $O/trace
unset APGPU_LIBRARY
AB_ARGS="" bash tools/ab_variants.sh 4 bail8 strip1 strip2 2>&1 | tee $O/ab_bench.txt
