#!/bin/bash
O=gpurun_out/r05m; mkdir -p $O
python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|Error|FAILED|assert" | tail -6 > $O/pytest_all.txt
cat $O/pytest_all.txt
python tools/bench_findbadpix.py > $O/findbadpix.txt 2>/dev/null; cat $O/findbadpix.txt
python tools/bench_kernels.py > $O/bench_kernels.txt 2>&1; grep -iE "box|flat_norm|background|A6|mesh" $O/bench_kernels.txt | cut -c1-160
python tools/bench_f32_sizes.py --u16 > $O/bench_u16_sizes.txt 2>&1; grep -E "N= ?(9[0-9]|1[0-2][0-9]) |sizes costing" $O/bench_u16_sizes.txt | cut -c1-130 | head -50
python bench.py > $O/bench_c2.json 2>/dev/null; cut -c1-400 $O/bench_c2.json
