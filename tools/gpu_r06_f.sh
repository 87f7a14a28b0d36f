#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r06; mkdir -p $O; rm -f $O/ablate_v2b.txt
timeout 900 python -m pytest tests/test_gpu_resample_stack.py -q -x 2>&1 | tail -5
timeout 600 python tools/bench_fused.py 2>&1 | grep "N=\|equal" > $O/bench_fused_v2b.txt; cat $O/bench_fused_v2b.txt
for v in fab1 fab2 fab8 fab15; do
  export APGPU_LIBRARY=$PWD/build_variants/$v/libapgpu.so
  echo "$v: $(timeout 300 python tools/bench_fused.py --fusedonly 2>&1 | grep 'N=' | tail -1)" >> $O/ablate_v2b.txt
done
cat $O/ablate_v2b.txt
