#!/usr/bin/env python3
"""Development aid: plans which compare-exchanges of the pruned 64-slot network run as 'hybrid' compare-exchanges - one operand
resident in LDS (ds_min_rtn_f32 / ds_max_rtn_f32 leaves one output there and returns the old value), the other in a register
(one v_max / v_min on the returned value) - and where operands move between registers and LDS (ds_read / ds_write: no VALU).
A hybrid compare-exchange costs 1 VALU + 1 DS instruction instead of 2 VALU.  Minimises  VALU + lambda * DS  with a MILP.
usage: plan_hybrid.py NP P0 T lambda [time_limit] -> prints a C++ op list (see tools/hybrid/lds_sort_bench.hip)."""
import sys
import numpy as np
from scipy.optimize import milp, LinearConstraint, Bounds


def batcher(P2, p0=1):
    ces = []
    p = p0
    while p < P2:
        k = p
        while k >= 1:
            j = k % p
            while j <= P2 - 1 - k:
                lim = min(k - 1, P2 - j - k - 1)
                for i in range(lim + 1):
                    if (i + j) // (p * 2) == (i + j + k) // (p * 2):
                        ces.append((i + j, i + j + k, (p, k)))
                j += 2 * k
            k //= 2
        p *= 2
    return ces


def pruned(NP, p0, T):
    full = batcher(NP, p0)
    if T <= 0:
        return full
    live = set(list(range(T)) + list(range(NP - T, NP)) + list(range((NP - T - 1) // 2, (NP + T) // 2 + 1)))
    keep = []
    for (a, b, g) in reversed(full):
        if a in live or b in live:
            keep.append((a, b, g))
            live.add(a)
            live.add(b)
    return keep[::-1]


def plan(NP, P0, T, lam, tlimit=60, final_in_regs=True):
    ces = pruned(NP, P0, T)
    n = len(ces)
    # variables: ra_c, rb_c (residency of the lower / upper wire during c: 1 = LDS), then one z per continuity edge / init / final
    idx_a = lambda c: 2 * c
    idx_b = lambda c: 2 * c + 1
    last = {}
    edges = []          # (var1, var2): cost |x1 - x2|
    unary = np.zeros(2 * n)
    for c, (a, b, g) in enumerate(ces):
        for w, v in ((a, idx_a(c)), (b, idx_b(c))):
            if w in last:
                edges.append((last[w], v))
            else:
                unary[v] += lam              # starts in a register: a write if LDS-resident at its first use
            last[w] = v
    if final_in_regs:
        for w, v in last.items():
            unary[v] += lam
    m = len(edges)
    nv = 2 * n + m
    cost = np.concatenate([unary, lam * np.ones(m)])
    # VALU: 2 - h_c, h_c = ra + rb (<= 1); DS for the hybrid op itself: lam * h_c
    for c in range(n):
        cost[idx_a(c)] += lam - 1
        cost[idx_b(c)] += lam - 1
    rows, lb, ub = [], [], []
    for c in range(n):
        r = np.zeros(nv); r[idx_a(c)] = 1; r[idx_b(c)] = 1
        rows.append(r); lb.append(0); ub.append(1)
    for e, (v1, v2) in enumerate(edges):
        for s in (1, -1):
            r = np.zeros(nv); r[2 * n + e] = 1; r[v1] = -s; r[v2] = s
            rows.append(r); lb.append(0); ub.append(np.inf)
    integ = np.zeros(nv); integ[:2 * n] = 1
    res = milp(cost, constraints=LinearConstraint(np.array(rows), lb, ub), integrality=integ, bounds=Bounds(0, 1), options={'time_limit': tlimit})
    x = np.round(res.x[:2 * n]).astype(int)
    return ces, x


def emit(NP, ces, x, out):
    """Op list in network order, group by group: moves, hybrid issues, plain compare-exchanges, hybrid completions."""
    n = len(ces)
    res = {w: 0 for w in range(NP)}                          # current residency
    groups = []
    for c, (a, b, g) in enumerate(ces):
        if not groups or groups[-1][0] != g:
            groups.append((g, []))
        groups[-1][1].append(c)
    nvalu = nds = 0
    tmp = 0
    row = {}
    free = []
    nrows = [0]

    def alloc(w):
        if free:
            row[w] = free.pop(0)
        else:
            row[w] = nrows[0]
            nrows[0] += 1
        return row[w]

    def release(w):
        free.append(row.pop(w))
        free.sort()
    for g, cs in groups:
        moves, issues, plains, dones = [], [], [], []
        for c in cs:
            a, b, _ = ces[c]
            ra, rb = x[2 * c], x[2 * c + 1]
            for w, r in ((a, ra), (b, rb)):
                if res[w] != r:
                    if r:
                        moves.append('H_W(%d, %d)' % (w, alloc(w)))
                    else:
                        moves.append('H_R(%d, %d)' % (w, row[w]))
                        release(w)
                    res[w] = r
                    nds += 1
            if ra:
                issues.append('H_LO_ISSUE(%d, %d, %d, t%d)' % (a, b, row[a], tmp)); dones.append('H_LO_DONE(%d, %d, t%d)' % (a, b, tmp)); tmp += 1
                nvalu += 1; nds += 1
            elif rb:
                issues.append('H_HI_ISSUE(%d, %d, %d, t%d)' % (a, b, row[b], tmp)); dones.append('H_HI_DONE(%d, %d, t%d)' % (a, b, tmp)); tmp += 1
                nvalu += 1; nds += 1
            else:
                plains.append('H_CE(%d, %d)' % (a, b))
                nvalu += 2
        out.write('    // merge level p = %d, k = %d\n' % g)
        for ops in (moves, issues, plains, dones):
            if ops:
                out.write('    ' + ' '.join(ops) + '\n')
    fin = []
    for w in range(NP):
        if res[w]:
            fin.append('H_R(%d, %d)' % (w, row[w]))
            nds += 1
    if fin:
        out.write('    // back to registers\n    ' + ' '.join(fin) + '\n')
    return nvalu, nds, nrows[0]


if __name__ == '__main__':
    NP, P0, T = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
    lam = float(sys.argv[4])
    tl = float(sys.argv[5]) if len(sys.argv) > 5 else 60
    ces, x = plan(NP, P0, T, lam, tl)
    import io
    buf = io.StringIO()
    nvalu, nds, nrows = emit(NP, ces, x, buf)
    print('// generated by tools/hybrid/plan_hybrid.py %s: %d compare-exchanges, %d VALU + %d DS instructions, %d LDS rows' % (' '.join(sys.argv[1:]), len(ces), nvalu, nds, nrows))
    sys.stdout.write(buf.getvalue())
