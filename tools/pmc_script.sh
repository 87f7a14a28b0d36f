#!/bin/bash
# SQ counters of one kernel under any python script (development aid):
#   bash tools/pmc_script.sh <tag> <kernel-substring> <script.py> [args...]
TAG=$1; KERN=$2; shift; shift
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/pmcs_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
SETS=("SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU"
      "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE"
      "SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_LEVEL_WAVES SQ_INSTS_SMEM"
      "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum TCP_PENDING_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum")
i=0
for S in "${SETS[@]}"; do
  i=$((i+1))
  timeout 180 rocprofv3 --pmc $S --output-format csv -d $OUT/p$i -- python3 $REPO/"$@" > $OUT/b$i.txt 2> $OUT/p$i.log
done
python3 - <<PY
import csv,glob,collections
for d in ('p1','p2','p3','p4'):
    acc=collections.defaultdict(list)
    for f in glob.glob('$OUT/%s/**/*counter_collection.csv'%d, recursive=True):
        for r in csv.DictReader(open(f)):
            if '$KERN' in r['Kernel_Name']:
                acc[r['Counter_Name']].append(float(r['Counter_Value']))
    for k,v in sorted(acc.items()): print(d,k,sum(v)/len(v), len(v))
    if not acc: print(open('$OUT/%s.log'%d).read()[-600:])
PY
