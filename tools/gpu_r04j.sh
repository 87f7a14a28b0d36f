#!/bin/bash
# one-off measurement script of round 4 (binary-search mad_std): run through gpurun
export APGPU_LIBRARY=$PWD/build_variants/mad/libapgpu.so
python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_classes.py tests/test_gpu_f64.py -x -q 2>&1 | grep -E "passed|failed|Error|FAILED|assert" | tail -8
python tools/bench_kernels.py 2>/dev/null | grep -E "ccdproc|EXTRA|fused, mean|plain" 
unset APGPU_LIBRARY
python tools/bench_kernels.py 2>/dev/null | grep -E "ccdproc|EXTRA|fused, mean|plain"
