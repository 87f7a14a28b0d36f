#!/bin/bash
O=gpurun_out/r05p; mkdir -p $O; rm -f $O/ab_resample.txt
R=$PWD
for v in prod nox phase0 prod nox phase0; do
  if [ $v = prod ]; then unset APGPU_LIBRARY; else export APGPU_LIBRARY=$R/build_variants/$v/libapgpu.so; fi
  for extra in "" "--scale 1.0001"; do
    echo -n "$v $extra: " >> $O/ab_resample.txt
    python tools/bench_resample.py --frames 16 --size 8192 --reps 9 $extra 2>/dev/null >> $O/ab_resample.txt
  done
done
cat $O/ab_resample.txt
