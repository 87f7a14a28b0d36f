#!/bin/bash
# round 5, first GPU pass: the new two-kernel plumbing (workspace, guard, unified redo pass) - tests, bench line, sweep
R=$PWD
O=gpurun_out/r05a; mkdir -p $O
python -m pytest tests/test_gpu_redo.py tests/test_gpu_parity.py tests/test_gpu_fuzz.py -x -q 2>&1 | tail -15 > $O/pytest_first.txt
cat $O/pytest_first.txt
python bench.py --steps 20 > $O/bench_line.json 2> $O/bench_line.log
cut -c1-1500 $O/bench_line.json
python tools/redo_sweep.py > $O/redo_sweep.txt 2> $O/redo_sweep.log
cat $O/redo_sweep.txt; tail -3 $O/redo_sweep.log
python -m pytest tests -m gpu -q 2>&1 | tail -15 > $O/pytest_all.txt
cat $O/pytest_all.txt
