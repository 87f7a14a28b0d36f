#!/bin/bash
# Instruction-cache and stall counters of one kernel (development aid): bash tools/pmc_icache.sh <tag> <kernel-substring> <bench args...>
TAG=$1; KERN=$2; shift; shift
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/pmci_$TAG
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_WAVES SQ_IFETCH SQ_WAIT_INST_ANY SQ_WAVE_CYCLES --output-format csv -d $OUT/p1 -- python3 $REPO/bench.py --steps 2 --warmup 1 --no-cpu-baseline "$@" > $OUT/b1.json 2> $OUT/p1.log
python3 - <<PY
import csv,glob,collections
acc=collections.defaultdict(list)
for f in glob.glob('$OUT/p1/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if '$KERN' in r['Kernel_Name']:
            acc[r['Counter_Name']].append(float(r['Counter_Value']))
for k,v in sorted(acc.items()): print(k,sum(v)/len(v), len(v))
if not acc: print(open('$OUT/p1.log').read()[-600:])
PY
rm -rf $OUT/p1
