#!/bin/bash
O=gpurun_out/r05h; mkdir -p $O
R=$PWD
python -m pytest tests/test_gpu_redo.py tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_bench_contract.py tests/test_gpu_fullsize.py -x -q 2>&1 | tail -12 > $O/pytest_a.txt
cat $O/pytest_a.txt
python tools/redo_sweep.py > $O/redo_sweep.txt 2> $O/redo_sweep.log; cat $O/redo_sweep.txt; tail -2 $O/redo_sweep.log
AB_ARGS="" bash tools/ab_variants.sh 4 prod nobail strip1 strip2 2>&1 | tee $O/ab_bench.txt
python bench.py --workload c5 --no-cpu-baseline > $O/bench_c5.json 2>/dev/null; python -c "
import json; d=json.load(open('$O/bench_c5.json')); print('C5', d['ms_per_step'], d['redo'])"
for v in prod onecopy norolling prod onecopy norolling; do
  if [ $v = prod ]; then unset APGPU_LIBRARY; else export APGPU_LIBRARY=$R/build_variants/$v/libapgpu.so; fi
  echo -n "$v: " >> $O/ab_resample.txt
  python tools/bench_resample.py --frames 16 --size 8192 --rot 0.2 >> $O/ab_resample.txt 2>> $O/ab.log
done
export APGPU_LIBRARY=$R/build_variants/onecopy/libapgpu.so
python -m pytest tests/test_gpu_resample.py tests/test_gpu_fullsize.py::test_c5_share_full_size_resample_clip -x -q 2>&1 | tail -5 >> $O/ab_resample.txt
unset APGPU_LIBRARY
cat $O/ab_resample.txt
