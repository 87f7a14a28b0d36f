#!/bin/bash
# round 6, second GPU call: full GPU suite on the rebuilt library (packed moments, G12 forms, fused resample + clip), the fused C5
# timing, and the MODE.IEEE experiment
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r06; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_resample_stack.py -q -x 2>&1 | tail -25 > $O/pytest_fused.txt
cat $O/pytest_fused.txt
timeout 900 python -m pytest tests -m gpu -q --deselect tests/test_gpu_resample_stack.py 2>&1 | tail -15 > $O/pytest_gpu_tail.txt
cat $O/pytest_gpu_tail.txt
timeout 600 python tools/bench_fused.py > $O/bench_fused.txt 2>&1; cat $O/bench_fused.txt
timeout 600 python tools/bench_fused.py --nomask >> $O/bench_fused.txt 2>&1; tail -3 $O/bench_fused.txt
timeout 300 tools/mfma_coissue > $O/mfma_coissue2.txt 2>&1; head -12 $O/mfma_coissue2.txt
timeout 900 bash tools/ab_variants.sh 5 prod noieee > $O/ab_noieee.txt 2>&1; cat $O/ab_noieee.txt
