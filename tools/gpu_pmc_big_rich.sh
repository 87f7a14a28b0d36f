#!/bin/bash
# As tools/gpu_pmc_big.sh, for the clipped mean with its median / std planes (tools/bench_big_rich.py NS=256): -> gpurun_out/pmc_big_rich.txt
R=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp
for ctr in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmcr_$ctr
  NS=256 rocprofv3 --pmc $ctr --output-format csv -d /tmp/pmcr_$ctr -- python3 $R/tools/bench_big_rich.py > /dev/null 2>&1
done
python3 - <<'PY'
import csv, glob, collections
tot = collections.defaultdict(lambda: collections.defaultdict(list))
for ctr in ('FETCH_SIZE', 'WRITE_SIZE'):
    for f in glob.glob('/tmp/pmcr_%s/**/*counter_collection.csv' % ctr, recursive=True):
        for r in csv.DictReader(open(f)):
            n = r['Kernel_Name']
            if 'stack_chunks_kernel' in n or 'std_pass' in n:
                tot[n.split('(')[0][-70:]][r['Counter_Name']].append(float(r['Counter_Value']))
P = 4096 * 4096
print('256 x 4096^2 float32 frames, fused calibration (tools/bench_big_rich.py NS=256): HBM bytes per launch (median over the dispatches; the')
print('chunk kernel is launched with lean outputs - mean, count: 268 B per pixel - and with mean, median, std, count - 284 B; the median of both is shown)')
for n, c in sorted(tot.items()):
    fs, ws = sorted(c.get('FETCH_SIZE', [0])), sorted(c.get('WRITE_SIZE', [0]))
    f, w = fs[len(fs) // 2] * 1024 * 2, ws[len(ws) // 2] * 1024
    alg = (256 * 4 + 12 + (12 + 4 if 'std_pass' in n else 8 + 4 + 12)) * P
    print('%-72s read %.3f GB written %.3f GB = %.3f x the algorithmic %.3f GB (%d dispatches)' % (n, f / 1e9, w / 1e9, (f + w) / alg, alg / 1e9, len(fs)))
PY
