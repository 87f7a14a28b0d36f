#!/usr/bin/env python3
"""Condenses a rocprofv3 output directory (profiles/run_profile.sh) into small text/JSON summaries."""
import csv
import glob
import json
import os
import sys

out, tag = sys.argv[1], sys.argv[2]
res = {'tag': tag}


def find(pattern):
    return sorted(glob.glob(os.path.join(out, pattern), recursive=True))


lines = []
for f in find('trace/**/*kernel_stats.csv'):
    rows = list(csv.DictReader(open(f)))
    for r in rows[:12]:
        r['Name'] = r.get('Name', '')[:100]
        lines.append(r)
res['kernel_stats'] = lines
durs = []
regs = None
for f in find('trace/**/*kernel_trace.csv'):
    for r in csv.DictReader(open(f)):
        if 'stack_sigclip' in r.get('Kernel_Name', ''):
            durs.append(int(r['End_Timestamp']) - int(r['Start_Timestamp']))
            regs = {k: r.get(k) for k in ('VGPR_Count', 'Accum_VGPR_Count', 'SGPR_Count', 'Scratch_Size', 'LDS_Block_Size',
                                          'Workgroup_Size', 'Grid_Size')}
if durs:
    res['stack_kernel'] = {'dispatches': len(durs), 'avg_ns': sum(durs) / len(durs), 'min_ns': min(durs), 'max_ns': max(durs),
                           'resources': regs}
for name, pat in (('FETCH_SIZE', 'pmc_fetch/**/*counter_collection.csv'), ('WRITE_SIZE', 'pmc_write/**/*counter_collection.csv')):
    vals = []
    for f in find(pat):
        for r in csv.DictReader(open(f)):
            if 'stack_sigclip' in r.get('Kernel_Name', '') and r.get('Counter_Name') == name:
                vals.append(float(r['Counter_Value']))
    if vals:
        res[name] = {'dispatches': len(vals), 'avg_raw': sum(vals) / len(vals)}
if 'FETCH_SIZE' in res and 'WRITE_SIZE' in res:
    # rocprofv3 reports KiB; gfx950 FETCH_SIZE counts 64 B per 128-B request of a wide coalesced stream (x2)
    fetch_b = res['FETCH_SIZE']['avg_raw'] * 1024 * 2
    write_b = res['WRITE_SIZE']['avg_raw'] * 1024
    res['hbm_bytes_per_launch'] = fetch_b + write_b
    res['hbm_read_bytes_corrected'] = fetch_b
    res['hbm_write_bytes'] = write_b
json.dump(res, open(os.path.join(out, 'summary_%s.json' % tag), 'w'), indent=1)
print(json.dumps({k: v for k, v in res.items() if k != 'kernel_stats'}, indent=1))
for r in lines[:8]:
    print(r)
