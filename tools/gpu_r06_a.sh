#!/bin/bash
# round 6, first GPU call: the two C2 experiments of VERDICT r5 next #1 (matrix-pipe co-issue microbenchmark; packed moments A/B)
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out/r06
timeout 300 tools/mfma_coissue > gpurun_out/r06/mfma_coissue.txt 2>&1
timeout 900 bash tools/ab_variants.sh 5 prod packed > gpurun_out/r06/ab_packed.txt 2>&1
tail -40 gpurun_out/r06/mfma_coissue.txt; cat gpurun_out/r06/ab_packed.txt
