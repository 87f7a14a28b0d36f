#!/bin/bash
# HBM traffic of the 129..512-frame kernels against their algorithmic bytes (through gpurun): tools/gpu_pmc_big.sh -> gpurun_out/pmc_big.txt
# Separate --pmc passes for FETCH_SIZE and WRITE_SIZE, nothing else traced (MI355X_MICROARCH.md); rocprofv3 reports KiB and gfx950's
# FETCH_SIZE counts 64 B per 128-B request of a coalesced stream (x 2) - the corrections of profiles/summarize.py.
R=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp
for ctr in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmcb_$ctr
  NS=256 rocprofv3 --pmc $ctr --output-format csv -d /tmp/pmcb_$ctr -- python3 $R/tools/bench_big_a6.py > /dev/null 2>&1
done
python3 - <<'PY'
import csv, glob, collections
tot = collections.defaultdict(lambda: collections.defaultdict(list))
for ctr in ('FETCH_SIZE', 'WRITE_SIZE'):
    for f in glob.glob('/tmp/pmcb_%s/**/*counter_collection.csv' % ctr, recursive=True):
        for r in csv.DictReader(open(f)):
            n = r['Kernel_Name']
            if 'rank_chunks' in n or 'mad_sums' in n:
                tot[n.split('(')[0][-70:]][r['Counter_Name']].append(float(r['Counter_Value']))
P = 4096 * 4096
print('256 x 4096^2 frames (tools/bench_big_a6.py NS=256): HBM bytes per launch from the counters against the algorithmic bytes')
for n, c in sorted(tot.items()):
    f = sorted(c.get('FETCH_SIZE', [0]))[len(c.get('FETCH_SIZE', [0])) // 2] * 1024 * 2
    w = sorted(c.get('WRITE_SIZE', [0]))[len(c.get('WRITE_SIZE', [0])) // 2] * 1024
    el = 2 if 'unsigned short' in n else 4
    mode = n.strip().rstrip('>').split(',')[-1].strip() if 'rank' in n else 'sums'
    alg = 256 * el * P + {'0': 8 * P, '1': 8 * P + 4 * P, '2': 4 * P, 'sums': 12 * P + 24 * P}.get(mode, 0)
    print('%-72s read %.3f GB written %.3f GB = %.3f x the algorithmic %.3f GB (%d dispatches)' % (n, f / 1e9, w / 1e9, (f + w) / alg, alg / 1e9, len(c.get('FETCH_SIZE', []))))
PY
