#!/bin/bash
O=gpurun_out/r05b; mkdir -p $O
python -m pytest tests/test_gpu_bench_contract.py::test_multi_rank_paths_on_one_gpu tests/test_gpu_fullsize.py::test_c5_share_full_size_resample_clip -x -q 2>&1 | tail -60 > $O/pytest_fail.txt
cat $O/pytest_fail.txt
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/trace -- python3 $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 2 --no-cpu-baseline > $GRAFT_REPO_ROOT/$O/bench_trace.json 2> $GRAFT_REPO_ROOT/$O/trace.log
cd $GRAFT_REPO_ROOT
f=$(find $O/trace -name "*kernel_stats.csv" | head -1); head -8 $f
t=$(find $O/trace -name "*kernel_trace.csv" | head -1)
python3 - $t <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
rows=[r for r in rows if 'stack' in r['Kernel_Name']]
rows.sort(key=lambda r:int(r['Start_Timestamp']))
prev=None
for r in rows[-12:]:
    s,e=int(r['Start_Timestamp']),int(r['End_Timestamp'])
    print('%-60s dur %8.1f us  gap-from-prev-end %8.1f us'%(r['Kernel_Name'][:60],(e-s)/1e3,(s-prev)/1e3 if prev else 0))
    prev=e
PY
rm -rf $O/trace
