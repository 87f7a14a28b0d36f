#!/bin/bash
# quick kernel-time check on the GPU box: bash tools/qbench.sh [bench args]
python bench.py --steps 20 --warmup 3 --no-cpu-baseline "$@" 2>&1 | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('ms/step %.4f  kernel avg %.4f min %.4f  frac %.3f  value %.0f' % (d['ms_per_step'], r['avg_launch_ms'], r['min_launch_ms'], r['frac'], d['value']))"
