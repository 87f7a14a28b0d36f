#!/bin/bash
# Per-kernel times of a python target (development aid): gpurun -- 'bash tools/prof_py.sh tools/prof_f4.py [args]'
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$REPO/gpurun_out/prof_py
rm -rf $OUT; mkdir -p $OUT
T=$REPO/$1; shift
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/t -- python3 $T "$@" > $OUT/run.log 2>&1
tail -8 $OUT/run.log
cd $REPO
python3 - <<'PY'
import csv,glob
f=glob.glob('gpurun_out/prof_py/t/*/*kernel_stats.csv')[0]
for i,r in enumerate(csv.DictReader(open(f))):
    if i>=28: break
    n=r['Name'].replace('(anonymous namespace)::','').replace('void ','')
    print(f"{n[:70]:70s} calls {r['Calls']:>5s} avg {float(r['AverageNs'])/1000:9.1f} us  total {float(r['TotalDurationNs'])/1000:10.1f}  {r['Percentage']}%")
PY
