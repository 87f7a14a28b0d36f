#!/bin/bash
# fast kernel for 72 .. 128 slots (variant library): parity, size sweep of the upper half
R=$PWD
mkdir -p gpurun_out/r04m
export APGPU_LIBRARY=$R/build_variants/wide/libapgpu.so
python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_classes.py -x -q 2>&1 | grep -E "passed|failed|Error|FAILED|assert" | tail -6
python tools/bench_f32_sizes.py > gpurun_out/r04m/bench_f32_sizes.txt 2>&1
grep -E "^N= *(6[5-9]|[7-9][0-9]|1[0-9][0-9])" gpurun_out/r04m/bench_f32_sizes.txt | cut -c1-60,100-140
tail -1 gpurun_out/r04m/bench_f32_sizes.txt
