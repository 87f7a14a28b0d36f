#!/usr/bin/env python3
"""Development aid: shrinks the merge phase of the 64-slot network for the outputs the float32 fast path reads
(clip_fast32: the 4 lowest, the 4 highest and ranks 29..34 in order; everything else only has to survive somewhere).
Start: Batcher's merge levels p = 16, 32 over four sorted 16-blocks, pruned by backward liveness (what make_pruned_net does).
Then compare-exchanges are removed greedily (and by random restarts) as long as the network stays correct on EVERY 0-1 input
whose 16-blocks are sorted (17^4 = 83521 inputs: exhaustive for the merge phase, by the 0-1 principle for selection)."""
import sys
import numpy as np

NP, T = 64, 4
NEED = list(range(T)) + list(range((NP - T - 1) // 2, (NP + T) // 2 + 1)) + list(range(NP - T, NP))


def batcher(P2, p0):
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
                        ces.append((i + j, i + j + k))
                j += 2 * k
            k //= 2
        p *= 2
    return ces


def liveness(ces, need):
    live = set(need)
    keep = []
    for (a, b) in reversed(ces):
        if a in live or b in live:
            keep.append((a, b))
            live.add(a)
            live.add(b)
    return keep[::-1]


def inputs(block):
    """All 0-1 inputs whose blocks of `block` wires are sorted ascending (zeros first), as a bool array [wires, cases]."""
    nb = NP // block
    counts = np.stack(np.meshgrid(*[np.arange(block + 1)] * nb, indexing='ij'), -1).reshape(-1, nb)    # ones per block
    x = np.zeros((NP, counts.shape[0]), bool)
    for b in range(nb):
        for i in range(block):
            x[b * block + i] = i >= block - counts[:, b]
    return x


def correct(ces, x, need):
    w = x.copy()
    for (a, b) in ces:
        lo = w[a] & w[b]
        hi = w[a] | w[b]
        w[a], w[b] = lo, hi
    ones = x.sum(0)                                          # number of ones: sorted output has ones in the top `ones` wires
    for j in need:
        if not np.array_equal(w[j], j >= NP - ones):
            return False
    return True


def main():
    block = int(sys.argv[1]) if len(sys.argv) > 1 else 16
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
    x = inputs(block)
    full = batcher(NP, block)
    ces = liveness(full, NEED)
    assert correct(ces, x, NEED)
    print('block', block, 'merge phase: full', len(full), 'liveness-pruned', len(ces), 'cases', x.shape[1])
    best = list(ces)
    for attempt in range(int(sys.argv[3]) if len(sys.argv) > 3 else 8):
        cur = list(ces)
        order = rng.permutation(len(cur)) if attempt else np.arange(len(cur))[::-1]
        removed = True
        while removed:
            removed = False
            for idx in sorted(order, key=lambda i: rng.random()) if attempt else list(order):
                if idx >= len(cur):
                    continue
                trial = cur[:idx] + cur[idx + 1:]
                if correct(trial, x, NEED):
                    cur = trial
                    removed = True
        print('attempt', attempt, len(cur))
        if len(cur) < len(best):
            best = cur
    print('best', len(best))
    print(best)


if __name__ == '__main__':
    main()
