#!/bin/bash
O=gpurun_out/r05d; mkdir -p $O
R=$PWD
python -m pytest tests/test_gpu_redo.py tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_bench_contract.py -x -q 2>&1 | tail -30 > $O/pytest_a.txt
cat $O/pytest_a.txt
python -m pytest tests/test_gpu_fullsize.py -x -q 2>&1 | tail -30 > $O/pytest_full.txt
cat $O/pytest_full.txt
python tools/redo_sweep.py > $O/redo_sweep.txt 2> $O/redo_sweep.log; cat $O/redo_sweep.txt; tail -2 $O/redo_sweep.log
AB_ARGS="" bash tools/ab_variants.sh 4 prod nobail strip1 strip2 2>&1 | tee $O/ab_bench.txt
python bench.py --workload c5 --no-cpu-baseline > $O/bench_c5.json 2>/dev/null; cut -c1-600 $O/bench_c5.json
bash tools/pmc_script.sh rolling resample_affine tools/bench_resample.py --frames 16 --size 8192 --reps 3 > $O/pmc_rolling.txt 2>&1
export APGPU_LIBRARY=$R/build_variants/norolling/libapgpu.so
bash tools/pmc_script.sh norolling resample_affine tools/bench_resample.py --frames 16 --size 8192 --reps 3 > $O/pmc_norolling.txt 2>&1
unset APGPU_LIBRARY
paste $O/pmc_rolling.txt $O/pmc_norolling.txt | cut -c1-200
