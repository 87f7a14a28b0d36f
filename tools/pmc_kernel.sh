#!/bin/bash
# SQ counters of one kernel (development aid): bash tools/pmc_kernel.sh <tag> <kernel-substring> <bench args...>
TAG=$1; KERN=$2; shift; shift
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/pmck_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU --output-format csv -d $OUT/p1 -- python3 $REPO/bench.py --steps 2 --warmup 1 --no-cpu-baseline "$@" > $OUT/b1.json 2> $OUT/p1.log
rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE --output-format csv -d $OUT/p2 -- python3 $REPO/bench.py --steps 2 --warmup 1 --no-cpu-baseline "$@" > $OUT/b2.json 2> $OUT/p2.log
python3 - <<PY
import csv,glob,collections
for d in ('p1','p2'):
    acc=collections.defaultdict(list)
    for f in glob.glob('$OUT/%s/**/*counter_collection.csv'%d, recursive=True):
        for r in csv.DictReader(open(f)):
            if '$KERN' in r['Kernel_Name']:
                acc[r['Counter_Name']].append(float(r['Counter_Value']))
    for k,v in sorted(acc.items()): print(d,k,sum(v)/len(v), len(v))
    if not acc: print(open('$OUT/%s.log'%d).read()[-400:])
PY
