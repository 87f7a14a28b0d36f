#!/bin/bash
# round 5: same-box A/B of the redo plumbing's variants + the stripped fast kernels (DESIGN 4.1 floor table) + rolling-window resample
O=gpurun_out/r05c; mkdir -p $O
R=$PWD
python -m pytest tests/test_gpu_resample.py tests/test_gpu_redo.py tests/test_gpu_bench_contract.py -x -q 2>&1 | tail -30 > $O/pytest_a.txt
cat $O/pytest_a.txt
python -m pytest tests/test_gpu_fullsize.py::test_c5_share_full_size_resample_clip -x -q 2>&1 | tail -40 > $O/pytest_c5.txt
cat $O/pytest_c5.txt
for rot in 0.2 1.0 3.0; do
for v in prod norolling prod norolling; do
  if [ $v = prod ]; then unset APGPU_LIBRARY; else export APGPU_LIBRARY=$R/build_variants/$v/libapgpu.so; fi
  echo -n "$v rot $rot: " >> $O/ab_resample.txt
  python tools/bench_resample.py --frames 16 --size 8192 --rot $rot >> $O/ab_resample.txt 2>> $O/ab.log
done
done
unset APGPU_LIBRARY
cat $O/ab_resample.txt
for v in prod cap3 nobail; do
  if [ $v = prod ]; then unset APGPU_LIBRARY; else export APGPU_LIBRARY=$R/build_variants/$v/libapgpu.so; fi
  echo "== $v" >> $O/ab.txt
  python tools/redo_sweep.py --rounds 1 --steps 8 --cases natural,100,"1 %" >> $O/ab.txt 2>> $O/ab.log
done
unset APGPU_LIBRARY
cat $O/ab.txt
AB_ARGS="" bash tools/ab_variants.sh 3 prod nobail cap3 strip1 strip2 2>&1 | tee $O/ab_bench.txt
python bench.py --workload c5 --no-cpu-baseline > $O/bench_c5.json 2>/dev/null; cut -c1-400 $O/bench_c5.json
python tools/cold_start.py > $O/cold_start.txt 2>&1; cat $O/cold_start.txt
