#!/bin/bash
# Development aid: parity tests that touch the global sigma clip, its bench line and its per-kernel times.
cd ${GRAFT_REPO_ROOT:-.}
python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_classes.py tests/test_gpu_background.py tests/test_gpu_lacosmic.py -m gpu -x -q -k "global or fuzz or badpix or background or lacosmic or pipeline" 2>&1 | tail -5
python tools/bench_kernels.py 2>/dev/null | grep -i global
bash tools/prof_global.sh > /dev/null 2>&1
python3 - <<'PY'
import csv,glob
f=glob.glob('gpurun_out/prof_global/t/*/*kernel_stats.csv')[0]
for r in csv.DictReader(open(f)):
    n=r['Name']
    if 'anonymous namespace)::' in n and ('GState' in n):
        short=n.split('::')[1].split('(')[0]
        print(f"{short:40s} calls {r['Calls']:>4s} avg {float(r['AverageNs'])/1000:8.1f} us  total {float(r['TotalDurationNs'])/1000:9.1f}")
PY
