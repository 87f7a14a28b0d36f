#!/bin/bash
# development aid: parity + kernel table on a variant library (through gpurun): tools/gpu_try.sh <variant>
R=$PWD
export APGPU_LIBRARY=$R/build_variants/$1/libapgpu.so
python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_classes.py tests/test_gpu_f64.py -x -q 2>&1 | grep -E "passed|failed|Error|FAILED|assert" | tail -6
for i in 1 2; do
python tools/bench_kernels.py 2>/dev/null | grep -E "ccdproc|EXTRA|fused, mean |plain"
done
unset APGPU_LIBRARY
python tools/bench_kernels.py 2>/dev/null | grep -E "ccdproc|EXTRA|fused, mean |plain"
