#!/bin/bash
O=gpurun_out/r05l; mkdir -p $O
R=$PWD
python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|Error|FAILED|assert" | tail -6 > $O/pytest_all.txt
cat $O/pytest_all.txt
for v in prod sgold prod sgold; do
  if [ $v = prod ]; then unset APGPU_LIBRARY; else export APGPU_LIBRARY=$R/build_variants/$v/libapgpu.so; fi
  echo "== $v" >> $O/findbadpix.txt
  python tools/bench_findbadpix.py >> $O/findbadpix.txt 2>/dev/null
done
unset APGPU_LIBRARY
cat $O/findbadpix.txt
python tools/bench_f32_sizes.py --u16 > $O/bench_u16_sizes.txt 2>&1; grep -E "N= ?(6[0-9]|7[0-9]|8[0-9]|9[0-9]|1[0-2][0-9]) |sizes costing" $O/bench_u16_sizes.txt | grep -E "\*|costing|<--" | cut -c1-130 | head -50
