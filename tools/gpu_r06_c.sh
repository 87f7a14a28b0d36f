#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r06; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -8 > $O/pytest_gpu_tail.txt
cat $O/pytest_gpu_tail.txt
timeout 900 python tools/bench_bands.py > $O/bench_bands.txt 2>&1; grep band $O/bench_bands.txt
timeout 1200 python tools/bench_frame_files.py --frames 8 > $O/frame_path.txt 2>&1; tail -12 $O/frame_path.txt
