#!/bin/bash
# measurement campaign of round 5 on the final library (full GPU suite, the four profiles, kernel table, size sweeps): run through gpurun
R=$PWD
O=gpurun_out/r05k
mkdir -p $O
python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|Error|FAILED|assert" | tail -6 > $O/pytest_gpu_tail.txt
cat $O/pytest_gpu_tail.txt
bash profiles/run_profile.sh r05 > /dev/null 2>&1
bash profiles/run_profile.sh r05_c5 --workload c5 > /dev/null 2>&1
bash profiles/run_profile.sh r05_c4 --workload c4 > /dev/null 2>&1
bash profiles/run_profile.sh r05_c3n1 --scaling strong --gpus 1 > /dev/null 2>&1
cd $R
python tools/bench_kernels.py > $O/bench_kernels.txt 2>&1
python tools/bench_f32_sizes.py > $O/bench_f32_sizes.txt 2>&1
python tools/bench_f32_sizes.py --u16 > $O/bench_u16_sizes.txt 2>&1
NS=512,448,384,300,257,256,192,129 python tools/bench_big.py 2>/dev/null > $O/bench_big.txt
python tools/bench_findbadpix.py 2>/dev/null > $O/bench_findbadpix.txt
python tools/cold_start.py 2>/dev/null > $O/cold_start.txt
python tools/redo_sweep.py 2>/dev/null > $O/redo_sweep.txt
tail -q -n 1 $O/bench_f32_sizes.txt $O/bench_u16_sizes.txt
for t in r05 r05_c5 r05_c4 r05_c3n1; do echo "== $t"; cat gpurun_out/prof_$t/bench_line.json | cut -c1-260; done
head -40 $O/bench_kernels.txt | cut -c1-120
cat $O/bench_big.txt | cut -c1-200
cat $O/redo_sweep.txt | cut -c1-160
python bench.py --force-collective --no-cpu-baseline --steps 10 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('force-collective (one rank):', d['ms_per_step'], 'ms', d.get('exchange'))"
