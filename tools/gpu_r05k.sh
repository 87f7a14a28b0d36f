#!/bin/bash
O=gpurun_out/r05k; mkdir -p $O
R=$PWD
python -m pytest tests -m gpu -x -q 2>&1 | tail -8 > $O/pytest_all.txt
cat $O/pytest_all.txt
python tools/bench_f32_sizes.py --u16 > $O/bench_u16_sizes.txt 2>&1; grep -E "N= ?(64|6[0-3]|7[0-9]|8[0-9]|9[0-9]|1[0-2][0-9]) |sizes costing" $O/bench_u16_sizes.txt | grep -E "\*|costing|<--" | cut -c1-130
python tools/bench_findbadpix.py > $O/findbadpix_prod.txt 2>&1; cat $O/findbadpix_prod.txt
export APGPU_LIBRARY=$R/build_variants/sgfold/libapgpu.so
python -m pytest tests/test_gpu_parity.py tests/test_gpu_classes.py tests/test_gpu_f64.py tests/test_gpu_background.py tests/test_gpu_lacosmic.py -x -q 2>&1 | tail -4 > $O/pytest_sgfold.txt
cat $O/pytest_sgfold.txt
python tools/bench_findbadpix.py > $O/findbadpix_sgfold.txt 2>&1; cat $O/findbadpix_sgfold.txt
unset APGPU_LIBRARY
python tools/bench_findbadpix.py >> $O/findbadpix_prod.txt 2>&1; tail -4 $O/findbadpix_prod.txt
python tools/bench_kernels.py > $O/bench_kernels.txt 2>&1; head -22 $O/bench_kernels.txt | cut -c1-110
