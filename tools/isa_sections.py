#!/usr/bin/env python3
"""Static instruction counts per APGPU_SECTION marker of one kernel in a .s file (development aid).
usage: isa_sections.py file.s kernel-name-substring"""
import collections
import re
import sys

txt = open(sys.argv[1]).read().split('\n')
want = sys.argv[2]
cur = None
sec = 'prologue'
counts = collections.OrderedDict()
for ln in txt:
    m = re.match(r'^(_ZN\S+):', ln)
    if m:
        cur = m.group(1) if want in m.group(1) else None
        sec = 'prologue'
        continue
    if cur is None:
        continue
    if 'APGPU_SECTION' in ln:
        sec = ln.split('APGPU_SECTION')[1].strip()
        continue
    m = re.match(r'^\s+([a-z][a-z0-9_]+)\s', ln + ' ')
    if m and not ln.strip().startswith(('.', ';')):
        op = m.group(1)
        c = counts.setdefault(sec, collections.Counter())
        kind = 'valu' if op.startswith('v_') else 'salu' if op.startswith('s_') else 'mem'
        c[kind] += 1
        c[op] += 1
    if 's_endpgm' in ln:
        cur = None
for sec, c in counts.items():
    top = [(k, v) for k, v in c.most_common(14) if k not in ('valu', 'salu', 'mem')]
    print('%-16s valu %5d salu %5d mem %4d | %s' % (sec, c['valu'], c['salu'], c['mem'], top[:10]))
