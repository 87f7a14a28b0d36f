#!/bin/bash
O=gpurun_out/r05o; mkdir -p $O
R=$PWD
python -m pytest tests/test_gpu_resample.py -m gpu -x -q 2>&1 | tail -3 | tee $O/pytest_resample.txt
for v in prod nokeep prod nokeep; do
  if [ $v = prod ]; then unset APGPU_LIBRARY; else export APGPU_LIBRARY=$R/build_variants/$v/libapgpu.so; fi
  echo -n "$v: " >> $O/ab_resample.txt
  python tools/bench_resample.py --frames 16 --size 8192 --reps 9 >> $O/ab_resample.txt 2>&1
  echo -n "$v 64x4096 os1: " >> $O/ab_resample.txt
  python tools/bench_resample.py --frames 64 --size 4096 --reps 9 >> $O/ab_resample.txt 2>&1
done
unset APGPU_LIBRARY
cat $O/ab_resample.txt
python bench.py --workload c5 > $O/bench_c5.json 2>/dev/null; cut -c1-300 $O/bench_c5.json
