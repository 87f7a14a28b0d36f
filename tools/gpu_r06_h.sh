#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r06; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -6 > $O/pytest_gpu_tail2.txt; cat $O/pytest_gpu_tail2.txt
timeout 1000 python tools/fuzz_long.py --minutes 12 --seed0 610000 --only fused,a6,resample,stack > $O/fuzz_fused_a6.txt 2>&1; tail -5 $O/fuzz_fused_a6.txt
