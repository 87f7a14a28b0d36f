#!/usr/bin/env python3
"""Basic blocks of one kernel in a .s file with VALU counts by issue class (development aid).
usage: isa_blocks.py file.s kernel-substring [min_valu]
Issue classes follow tools/issue_cost.hip (profiles/r03/issue_cost.txt): 'fast' = 2-cycle wave64 forms, 'slow' = 4-cycle."""
import collections
import re
import sys

FAST = {'v_add_f32', 'v_sub_f32', 'v_subrev_f32', 'v_mul_f32', 'v_mov_b32', 'v_and_b32', 'v_xor_b32', 'v_or_b32', 'v_add_u32', 'v_sub_u32',
        'v_subrev_u32', 'v_fmac_f32', 'v_lshrrev_b32', 'v_ashrrev_i32', 'v_max_u16', 'v_min_u16', 'v_not_b32', 'v_cvt_f32_u32', 'v_cvt_f32_i32',
        'v_fma_f32', 'v_mac_f32', 'v_fmaak_f32', 'v_fmamk_f32', 'v_accvgpr_write_b32', 'v_accvgpr_read_b32'}


def base(op):
    return re.sub(r'_(e32|e64|dpp|sdwa)$', '', op)


def main():
    txt = open(sys.argv[1]).read().split('\n')
    want = sys.argv[2]
    minv = int(sys.argv[3]) if len(sys.argv) > 3 else 0
    cur = None
    blocks = []
    blk = None
    for i, ln in enumerate(txt):
        m = re.match(r'^(_ZN\S+):', ln)
        if m:
            cur = m.group(1) if want in m.group(1) else None
            if cur:
                blk = {'name': 'entry', 'line': i, 'ops': collections.Counter(), 'term': [], 'sec': ''}
                blocks.append(blk)
            continue
        if cur is None:
            continue
        m = re.match(r'^(\.LBB\S+):', ln)
        if m:
            blk = {'name': m.group(1), 'line': i, 'ops': collections.Counter(), 'term': [], 'sec': ''}
            blocks.append(blk)
            continue
        if 'APGPU_SECTION' in ln:
            blk['sec'] += ' [' + ln.split('APGPU_SECTION')[1].strip() + ']'
            continue
        m = re.match(r'^\s+([a-z][a-z0-9_]+)\s*(.*)$', ln)
        if m and not ln.strip().startswith(('.', ';')):
            op = m.group(1)
            blk['ops'][op] += 1
            if op.startswith('s_cbranch') or op == 's_branch':
                blk['term'].append(op.replace('s_cbranch_', '') + '->' + m.group(2).split()[0])
            if op == 's_endpgm':
                blk['term'].append('END')
    for b in blocks:
        valu = sum(v for k, v in b['ops'].items() if k.startswith('v_'))
        if valu < minv:
            continue
        fast = sum(v for k, v in b['ops'].items() if k.startswith('v_') and base(k) in FAST)
        salu = sum(v for k, v in b['ops'].items() if k.startswith('s_'))
        mem = sum(v for k, v in b['ops'].items() if not k.startswith(('v_', 's_')))
        top = ', '.join('%s %d' % (k, v) for k, v in b['ops'].most_common(6) if k.startswith('v_'))
        print('%-10s L%-6d valu %4d (fast %4d slow %4d) salu %4d mem %3d %s | %s | %s' % (b['name'], b['line'], valu, fast, valu - fast, salu, mem, b['sec'], ' '.join(b['term']), top))


main()
