#!/bin/bash
O=gpurun_out/r05p; mkdir -p $O; rm -f $O/ab_resample.txt
R=$PWD
python -m pytest tests/test_gpu_resample.py -m gpu -x -q 2>&1 | tail -2
for v in prod reload prod reload; do
  if [ $v = prod ]; then unset APGPU_LIBRARY; else export APGPU_LIBRARY=$R/build_variants/$v/libapgpu.so; fi
  for extra in "" "--rot 0.1" "--rot 0.6" "--scale 1.0001" "--scale 1.001" "--rot 1.5"; do
    echo -n "$v $extra: " >> $O/ab_resample.txt
    python tools/bench_resample.py --frames 16 --size 8192 --reps 9 $extra 2>/dev/null >> $O/ab_resample.txt
  done
done
cat $O/ab_resample.txt
python bench.py --workload c5 > $O/bench_c5.json 2>/dev/null; cut -c1-300 $O/bench_c5.json
