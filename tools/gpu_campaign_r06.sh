#!/bin/bash
# measurement campaign of round 6 on the final library (full GPU suite, the profiles, kernel table, sweeps): run through gpurun
R=$PWD
O=gpurun_out/r06k
mkdir -p $O
python -m pytest tests -m gpu -q 2>&1 | grep -E "passed|failed|Error|FAILED|assert" | tail -6 > $O/pytest_gpu_tail.txt
cat $O/pytest_gpu_tail.txt
bash profiles/run_profile.sh r06 > /dev/null 2>&1
bash profiles/run_profile.sh r06_c5 --workload c5 > /dev/null 2>&1
bash profiles/run_profile.sh r06_c5_fused --workload c5 --fused > /dev/null 2>&1
bash profiles/run_profile.sh r06_c4 --workload c4 > /dev/null 2>&1
bash profiles/run_profile.sh r06_c3n1 --scaling strong --gpus 1 > /dev/null 2>&1
cd $R
python tools/bench_kernels.py > $O/bench_kernels.txt 2>&1
python tools/bench_a6.py > $O/bench_a6.txt 2>&1
python tools/bench_f32_sizes.py > $O/bench_f32_sizes.txt 2>&1
python tools/bench_f32_sizes.py --u16 > $O/bench_u16_sizes.txt 2>&1
NS=512,448,384,300,257,256,192,129 python tools/bench_big.py 2>/dev/null > $O/bench_big.txt
python tools/bench_fused.py 2>&1 | grep "N=\|equal" > $O/bench_fused.txt
python tools/cold_start.py 2>/dev/null > $O/cold_start.txt
tail -q -n 1 $O/bench_f32_sizes.txt $O/bench_u16_sizes.txt
for t in r06 r06_c5 r06_c5_fused r06_c4 r06_c3n1; do echo "== $t"; cat gpurun_out/prof_$t/bench_line.json | cut -c1-260; done
head -40 $O/bench_kernels.txt | cut -c1-120
cat $O/bench_big.txt | cut -c1-200
cat $O/bench_a6.txt | cut -c1-160
cat $O/bench_fused.txt
python bench.py --force-collective --no-cpu-baseline --steps 10 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('force-collective (one rank):', d['ms_per_step'], 'ms', d.get('exchange'), 'compute_ms', d.get('compute_ms'), 'overlap', d.get('exchange_overlap_frac'))"
python bench.py --force-collective --no-gather --no-cpu-baseline --steps 10 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('force-collective, no gather (one rank):', d['ms_per_step'], 'ms', d.get('exchange_bytes_on_wire'))"
