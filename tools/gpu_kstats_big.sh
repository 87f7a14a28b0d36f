#!/bin/bash
# rocprofv3 kernel table of the 129..512-frame paths (through gpurun): tools/gpu_kstats_big.sh  -> gpurun_out/kstats_big.txt
R=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp
for job in "a6 256,512" "rich 256,512"; do
  set -- $job
  rm -rf /tmp/ksb
  if [ $1 = a6 ]; then NS=$2 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ksb -- python3 $R/tools/bench_big_a6.py > /dev/null 2>&1
  else NS=$2 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ksb -- python3 $R/tools/bench_big_rich.py > /dev/null 2>&1; fi
  f=$(find /tmp/ksb -name "*kernel_stats.csv" | head -1)
  echo "== tools/bench_big_$1.py NS=$2 (4096^2; float32 and, for a6, uint16): kernel, calls, average / min / max us"
  python3 - $f <<'PY'
import csv, sys
for r in list(csv.DictReader(open(sys.argv[1]))):
    n = r['Name']
    if any(k in n for k in ('stack_', 'apgpu')):
        print('%-100s %4s calls  avg %10.1f  min %10.1f  max %10.1f us' % (n[:100], r['Calls'], float(r['AverageNs']) / 1e3, float(r['MinNs']) / 1e3, float(r['MaxNs']) / 1e3))
PY
done
