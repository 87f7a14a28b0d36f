#!/bin/bash
O=gpurun_out/r05j; mkdir -p $O
R=$PWD
python -m pytest tests/test_gpu_redo.py tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_bench_contract.py tests/test_gpu_fullsize.py -x -q 2>&1 | tail -8 > $O/pytest_a.txt
cat $O/pytest_a.txt
python tools/redo_sweep.py > $O/redo_sweep.txt 2> $O/redo_sweep.log; cat $O/redo_sweep.txt; tail -2 $O/redo_sweep.log
python bench.py --workload c5 --no-cpu-baseline > $O/bench_c5.json 2>/dev/null; python -c "
import json; d=json.load(open('$O/bench_c5.json')); print('C5', d['ms_per_step'], d['redo'])"
export APGPU_LIBRARY=$R/build_variants/w16/libapgpu.so
python -m pytest tests/test_gpu_resample.py tests/test_gpu_fullsize.py::test_c5_share_full_size_resample_clip tests/test_gpu_classes.py -x -q 2>&1 | tail -5 > $O/pytest_w16.txt
cat $O/pytest_w16.txt
for rot in 0.2 1.0 3.0; do
for v in prod w16 prod w16; do
  if [ $v = prod ]; then unset APGPU_LIBRARY; else export APGPU_LIBRARY=$R/build_variants/$v/libapgpu.so; fi
  echo -n "$v rot $rot: " >> $O/ab_resample.txt
  python tools/bench_resample.py --frames 16 --size 8192 --rot $rot >> $O/ab_resample.txt 2>> $O/ab.log
done
done
cat $O/ab_resample.txt
export APGPU_LIBRARY=$R/build_variants/w16/libapgpu.so
bash tools/pmc_script.sh w16 resample_affine tools/bench_resample.py --frames 16 --size 8192 --reps 3 > $O/pmc_w16.txt 2>&1
cat $O/pmc_w16.txt
