#!/usr/bin/env python3
"""Condenses one profiles/run_profile.sh output directory (all passes of ONE gpurun call) into a single summary:
summary_<tag>.json (+ kernel_stats_top.csv, dominant_kernel_dispatches.csv, pmc_dominant_kernel.csv) written next to
the raw rocprofv3 output; profiles/collect.py copies them into profiles/<tag>/."""
import csv
import glob
import json
import os
import sys

out, tag = sys.argv[1], sys.argv[2]
res = {'tag': tag}


def find(pattern):
    return sorted(glob.glob(os.path.join(out, pattern), recursive=True))


def load_line(name):
    try:
        txt = open(os.path.join(out, name)).read().strip().splitlines()
        return json.loads(txt[-1])
    except Exception:
        return None


line = load_line('bench_line.json')
traced = load_line('bench_trace.json')
res['bench_line'] = line
res['bench_line_under_rocprof'] = traced
kern = (line or traced or {}).get('roofline', {}).get('kernel', 'stack_sigclip')
named = kern.split(' + ')[-1].split(' (')[0]               # the library's own name of the dispatched variant


def is_library_kernel(name):
    # libapgpu.so's kernels live in apgpu_stack:: or in anonymous namespaces; PyTorch's (synthetic data, copies) do not
    return 'apgpu_stack::' in name or '(anonymous namespace)::' in name and 'at::native' not in name


# Every dispatch of the library in the traced run: the table lists ALL of them (a step of the stack is two dispatches, one of
# C5 five), and the DOMINANT kernel - the one the counters are condensed for - is the library kernel with the largest TOTAL
# duration (round 4 took the name in the bench line, which for C5 is the stack kernel while the resample takes 3.7 of 5.4 ms).
stats = []
for f in find('trace/**/*kernel_stats.csv'):
    rows = list(csv.DictReader(open(f)))
    lib = [r for r in rows if is_library_kernel(r.get('Name', ''))]
    other = [r for r in rows if not is_library_kernel(r.get('Name', ''))][:6]
    with open(os.path.join(out, 'kernel_stats_top.csv'), 'w', newline='') as fh:
        w = csv.writer(fh)
        cols = list(rows[0].keys()) if rows else []
        w.writerow(['library_kernel'] + cols)
        for r in lib + other:
            w.writerow([int(is_library_kernel(r['Name']))] + [r[c][:160] if c == 'Name' else r[c] for c in cols])
    for r in lib + other:
        r['Name'] = r.get('Name', '')[:160]
        stats.append(r)
res['kernel_stats'] = stats
lib_stats = [r for r in stats if is_library_kernel(r['Name'])]
key = named
if lib_stats:
    top = max(lib_stats, key=lambda r: float(r['TotalDurationNs']))
    # the trace prints 'void ns::kernel<...>(args)': keep what identifies the kernel in the other CSVs
    key = top['Name'].replace('void ', '').split('(apgpu_stack::StackParams)')[0]
    key = key.split('>(')[0] + '>' if '>(' in key else key
res['dominant_kernel'] = key
res['bench_line_kernel'] = named

durs, regs = [], None
keep = ['Kernel_Name', 'Start_Timestamp', 'End_Timestamp', 'VGPR_Count', 'Accum_VGPR_Count', 'SGPR_Count', 'Scratch_Size',
        'LDS_Block_Size', 'Workgroup_Size', 'Grid_Size']
with open(os.path.join(out, 'dominant_kernel_dispatches.csv'), 'w', newline='') as fh:
    w = csv.writer(fh)
    w.writerow(keep + ['Duration_ns'])
    for f in find('trace/**/*kernel_trace.csv'):
        for r in csv.DictReader(open(f)):
            if key in r.get('Kernel_Name', ''):
                d = int(r['End_Timestamp']) - int(r['Start_Timestamp'])
                durs.append(d)
                regs = {k: r.get(k) for k in keep[3:]}
                w.writerow([r.get(k, '')[:140] for k in keep] + [d])
if durs:
    res['dominant_kernel_trace'] = {'dispatches': len(durs), 'avg_ns': sum(durs) / len(durs), 'min_ns': min(durs),
                                    'max_ns': max(durs), 'rocprof_resources': regs,
                                    'note': 'rocprofv3 prints VGPR_Count = allocated VGPRs / 2 on gfx950 (the allocation granule of 8 '
                                            'registers reported in units of 4): VGPR_Count 80 = 160 allocated = the 153 registers of the '
                                            'code object metadata (profiles/<tag>/kernel_resources.csv) rounded up to the granule'}

counters = {}
with open(os.path.join(out, 'pmc_dominant_kernel.csv'), 'w', newline='') as fh:
    w = csv.writer(fh)
    w.writerow(['pass', 'Kernel_Name', 'Counter_Name', 'Counter_Value'])
    for f in find('pmc_*/**/*counter_collection.csv'):
        p = os.path.relpath(f, out).split(os.sep)[0]
        for r in csv.DictReader(open(f)):
            if key in r.get('Kernel_Name', ''):
                counters.setdefault(r['Counter_Name'], []).append(float(r['Counter_Value']))
                w.writerow([p, r['Kernel_Name'][:140], r['Counter_Name'], r['Counter_Value']])
avg = {k: sum(v) / len(v) for k, v in counters.items()}
res['pmc_avg_per_dispatch'] = avg
res['pmc_dispatches'] = {k: len(v) for k, v in counters.items()}
if 'FETCH_SIZE' in avg and 'WRITE_SIZE' in avg:
    # rocprofv3 reports KiB; gfx950 FETCH_SIZE counts 64 B per 128-B request of a wide coalesced stream (x2)
    fetch_b = avg['FETCH_SIZE'] * 1024 * 2
    write_b = avg['WRITE_SIZE'] * 1024
    res['hbm_bytes_per_launch'] = fetch_b + write_b
    res['hbm_read_bytes_corrected'] = fetch_b
    res['hbm_write_bytes'] = write_b
if 'SQ_INSTS_VALU' in avg and avg.get('SQ_WAVES'):
    res['valu_insts_per_wave'] = avg['SQ_INSTS_VALU'] / avg['SQ_WAVES']
    for k in ('SQ_INSTS_SALU', 'SQ_INSTS_VMEM_RD', 'SQ_INSTS_LDS'):
        if k in avg:
            res[k.lower().replace('sq_insts_', '') + '_insts_per_wave'] = avg[k] / avg['SQ_WAVES']
if 'GRBM_GUI_ACTIVE' in avg and durs:
    # rocprofv3 sums GRBM_GUI_ACTIVE over the 8 XCDs of the chip
    res["gpu_clock_ghz_during_kernel"] = avg["GRBM_GUI_ACTIVE"] / 8.0 / (sum(durs) / len(durs))
if 'SQ_ACTIVE_INST_VALU' in avg and 'GRBM_GUI_ACTIVE' in avg:
    # SQ_ACTIVE_INST_VALU counts in units of 4 cycles (one wave64 issue slot of a SIMD), summed over the chip's 1024 SIMDs;
    # GRBM_GUI_ACTIVE / 8 = the kernel's busy cycles per XCD
    res['valu_busy_frac'] = avg['SQ_ACTIVE_INST_VALU'] * 4.0 / (1024.0 * avg['GRBM_GUI_ACTIVE'] / 8.0)
    if res['valu_busy_frac'] > 1.0:                           # cannot be: numerator and denominator are not the same kernel's
        res['valu_busy_frac_rejected'] = res.pop('valu_busy_frac')
if 'SQ_WAVE_CYCLES' in avg and avg.get('SQ_BUSY_CYCLES'):
    res['sq_wave_cycles_over_busy_cycles'] = avg['SQ_WAVE_CYCLES'] / avg['SQ_BUSY_CYCLES']
json.dump(res, open(os.path.join(out, 'summary_%s.json' % tag), 'w'), indent=1)
print(json.dumps({k: v for k, v in res.items() if k not in ('kernel_stats', 'bench_line', 'bench_line_under_rocprof')}, indent=1))
for r in stats[:8]:
    print(r)
