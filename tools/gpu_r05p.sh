#!/bin/bash
O=gpurun_out/r05p; mkdir -p $O; rm -f $O/ab_resample.txt
R=$PWD
python -m pytest tests/test_gpu_resample.py -m gpu -x -q 2>&1 | tail -2
for v in prod w6 nokeep prod w6 nokeep; do
  if [ $v = prod ]; then unset APGPU_LIBRARY; else export APGPU_LIBRARY=$R/build_variants/$v/libapgpu.so; fi
  echo -n "$v: " >> $O/ab_resample.txt
  python tools/bench_resample.py --frames 16 --size 8192 --reps 9 2>/dev/null >> $O/ab_resample.txt
  echo -n "$v scale 1.001: " >> $O/ab_resample.txt
  python tools/bench_resample.py --frames 16 --size 8192 --reps 9 --scale 1.001 2>/dev/null >> $O/ab_resample.txt
  echo -n "$v rot 1.5: " >> $O/ab_resample.txt
  python tools/bench_resample.py --frames 16 --size 8192 --reps 9 --rot 1.5 2>/dev/null >> $O/ab_resample.txt
done
cat $O/ab_resample.txt
