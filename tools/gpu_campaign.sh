#!/bin/bash
# measurement campaign of a round on the final library (full GPU suite, the four profiles, kernel table, size sweeps): run through gpurun
R=$PWD
mkdir -p gpurun_out/r04k
python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|Error|FAILED|assert" | tail -6 > gpurun_out/r04k/pytest_gpu_tail.txt
cat gpurun_out/r04k/pytest_gpu_tail.txt
bash profiles/run_profile.sh r04 > /dev/null 2>&1
bash profiles/run_profile.sh r04_c5 --workload c5 > /dev/null 2>&1
bash profiles/run_profile.sh r04_c4 --workload c4 > /dev/null 2>&1
bash profiles/run_profile.sh r04_c3n1 --scaling strong --gpus 1 > /dev/null 2>&1
cd $R
python tools/bench_kernels.py > gpurun_out/r04k/bench_kernels.txt 2>&1
python tools/bench_f32_sizes.py > gpurun_out/r04k/bench_f32_sizes.txt 2>&1
python tools/bench_f32_sizes.py --u16 > gpurun_out/r04k/bench_u16_sizes.txt 2>&1
python tools/bench_u16_64.py 2>&1 | tail -2
tail -q -n 1 gpurun_out/r04k/bench_f32_sizes.txt gpurun_out/r04k/bench_u16_sizes.txt
grep -- "<--" gpurun_out/r04k/bench_f32_sizes.txt | cut -c1-12,100-140 | tr '\n' ';'
echo
grep -- "<--" gpurun_out/r04k/bench_u16_sizes.txt | cut -c1-12,100-140 | tr '\n' ';'
echo
for t in r04 r04_c5 r04_c4 r04_c3n1; do echo "== $t"; cat gpurun_out/prof_$t/bench_line.json | cut -c1-300; done
head -30 gpurun_out/r04k/bench_kernels.txt | cut -c1-120
python bench.py --force-collective --no-cpu-baseline --steps 10 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('force-collective (one rank, 4 stripes):', d['ms_per_step'], 'ms')"
