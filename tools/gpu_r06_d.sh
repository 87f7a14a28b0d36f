#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r06; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_resample_stack.py -q -x 2>&1 | tail -25 > $O/pytest_fused.txt
cat $O/pytest_fused.txt
timeout 600 python tools/bench_fused.py > $O/bench_fused_v2.txt 2>&1; grep -v "^RCCL\|^HIP\|^ROCm\|^Host\|^Librccl\|amdgpu.ids" $O/bench_fused_v2.txt
APGPU_FUSED_V1=1 timeout 600 python tools/bench_fused.py 2>&1 | grep "N=" | tail -1
timeout 600 python tools/bench_fused.py --nomask 2>&1 | grep "N=\|equal" | tail -2
