#!/bin/bash
# the randomised parity run on the GPU box (through gpurun): tools/gpu_fuzz.sh <minutes> <out file under gpurun_out/>
mkdir -p gpurun_out/$(dirname $2)
python -m pytest tests/test_gpu_fuzz.py tests/test_gpu_parity.py -x -q 2>&1 | grep -E "passed|failed|Error|FAILED|assert" | tail -5
python tools/fuzz_long.py --minutes $1 > gpurun_out/$2 2>&1
tail -15 gpurun_out/$2
