#!/bin/bash
O=gpurun_out/r05n; mkdir -p $O
python -m pytest tests/test_gpu_redo.py tests/test_gpu_parity.py tests/test_gpu_fuzz.py -m gpu -x -q 2>&1 | tail -15 > $O/pytest.txt
cat $O/pytest.txt
NS=512,448,384,300,257,256 python tools/bench_big.py > $O/bench_big.txt 2>&1; cat $O/bench_big.txt | cut -c1-250
