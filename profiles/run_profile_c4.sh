#!/bin/bash
# HBM traffic of the C4 kernel (uint16 Bayer median, full size): separate --pmc passes, as profiles/run_profile.sh does for C2.
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/prof_c4
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="--workload c4 --height 6248 --width 4176 --steps 3 --warmup 1 --no-cpu-baseline"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $REPO/bench.py $ARGS > $OUT/bench_fetch.json 2> $OUT/fetch.log
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $REPO/bench.py $ARGS > $OUT/bench_write.json 2> $OUT/write.log
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $REPO/bench.py --workload c4 --height 6248 --width 4176 --steps 10 --warmup 2 --no-cpu-baseline > $OUT/bench_trace.json 2> $OUT/trace.log
python3 - <<PY
import csv,glob,json
res={}
for name,pat in (('FETCH_SIZE','pmc_fetch'),('WRITE_SIZE','pmc_write')):
    v=[float(r['Counter_Value']) for f in glob.glob('$OUT/%s/*/*counter_collection.csv'%pat) for r in csv.DictReader(open(f)) if 'stack_median_u16' in r['Kernel_Name'] and r['Counter_Name']==name]
    res[name]=sum(v)/len(v) if v else None
d=[]
for f in glob.glob('$OUT/trace/*/*kernel_trace.csv'):
    for r in csv.DictReader(open(f)):
        if 'stack_median_u16' in r['Kernel_Name']: d.append(int(r['End_Timestamp'])-int(r['Start_Timestamp']))
out={'kernel':'stack_median_u16_kernel<64,calib,full>','workload':'C4 64x6248x4176 uint16','algorithmic_bytes':2*64*6248*4176+16*6248*4176,
     'hbm_read_bytes_fetch_size_x2':res['FETCH_SIZE']*1024*2 if res['FETCH_SIZE'] else None,'hbm_write_bytes':res['WRITE_SIZE']*1024 if res['WRITE_SIZE'] else None,
     'rocprof_avg_ns':sum(d)/len(d) if d else None,'rocprof_dispatches':len(d)}
json.dump(out,open('$OUT/pmc_c4.json','w'),indent=1); print(json.dumps(out,indent=1))
PY
