#!/bin/bash
# development aid: rocprofv3 kernel table (name, calls, average us) of a short bench run (through gpurun): tools/gpu_kstats.sh [bench args]
R=$PWD
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks -- python3 $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline "$@" > /dev/null 2>&1
f=$(find /tmp/ks -name "*kernel_stats.csv" | head -1)
python3 - $f <<'PY'
import csv, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:14]:
    print('%-90s %5s calls  %10.1f us' % (r['Name'][:90], r['Calls'], float(r['AverageNs']) / 1e3))
PY
