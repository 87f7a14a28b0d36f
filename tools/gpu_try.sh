#!/bin/bash
python -m pytest tests/test_gpu_redo.py tests/test_gpu_parity.py tests/test_gpu_classes.py tests/test_gpu_f64.py -m gpu -x -q 2>&1 | tail -12
python tools/bench_kernels.py 2>/dev/null | grep -E "A6|ccdproc" | cut -c1-140
