#!/bin/bash
# dynamic VALU instruction mix of the stack kernel (development aid): bash tools/pmc_mix.sh <tag> [bench args]
TAG=${1:-x}; shift
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/mix_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_INT32 --output-format csv -d $OUT/p1 -- python3 $REPO/bench.py --steps 3 --warmup 1 --no-cpu-baseline "$@" > $OUT/b1.json 2> $OUT/p1.log
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_INT64 SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_WAVES --output-format csv -d $OUT/p2 -- python3 $REPO/bench.py --steps 3 --warmup 1 --no-cpu-baseline "$@" > $OUT/b2.json 2> $OUT/p2.log
python3 - <<PY
import csv,glob,collections
tot={}
for d in ('p1','p2'):
    acc=collections.defaultdict(list)
    for f in glob.glob('$OUT/%s/**/*counter_collection.csv'%d, recursive=True):
        for r in csv.DictReader(open(f)):
            if 'stack_sigclip' in r['Kernel_Name']:
                acc[r['Counter_Name']].append(float(r['Counter_Value']))
    for k,v in acc.items(): tot[k]=sum(v)/len(v)
w=tot.get('SQ_WAVES',262144)
for k,v in sorted(tot.items()): print('%-28s %10.1f per wave' % (k, v/w))
PY
