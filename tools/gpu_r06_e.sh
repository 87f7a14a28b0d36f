#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r06; mkdir -p $O; rm -f $O/ablate_v2.txt
for v in prod fab1 fab2 fab8 fab15; do
  if [ "$v" = prod ]; then unset APGPU_LIBRARY; else export APGPU_LIBRARY=$PWD/build_variants/$v/libapgpu.so; fi
  echo "$v: $(timeout 300 python tools/bench_fused.py --fusedonly 2>&1 | grep 'N=' | tail -1)" >> $O/ablate_v2.txt
done
cat $O/ablate_v2.txt
