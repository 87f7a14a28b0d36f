#!/bin/bash
# static pads + uint16 pair fast kernel (variant library): parity, size sweeps
R=$PWD
mkdir -p gpurun_out/r04l
export APGPU_LIBRARY=$R/build_variants/pads/libapgpu.so
python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_classes.py -x -q 2>&1 | grep -E "passed|failed|Error|FAILED|assert" | tail -6
python tools/bench_f32_sizes.py > gpurun_out/r04l/bench_f32_sizes.txt 2>&1
python tools/bench_f32_sizes.py --u16 > gpurun_out/r04l/bench_u16_sizes.txt 2>&1
grep -E "^N= *(6[0-4]|5[0-9]|3[0-9]|2[0-9]|1[2-9])" gpurun_out/r04l/bench_f32_sizes.txt | cut -c1-60,100-140
tail -1 gpurun_out/r04l/bench_f32_sizes.txt
grep -E "^N= *(6[0-4]|5[0-9]|3[0-9]|2[0-9]|1[2-9])" gpurun_out/r04l/bench_u16_sizes.txt | cut -c1-60,100-140
tail -1 gpurun_out/r04l/bench_u16_sizes.txt
