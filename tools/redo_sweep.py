#!/usr/bin/env python3
"""The two-kernel scheme against its worst case: 64 x 4096 x 4096 float32, fused calibration + 3-sigma
clipped mean, with a chosen share of the pixel columns FORCED off the fast kernel - five frames of such a column carry a
+3000 ADU outlier, so the first pass trims five values from the high side, one more than the fast path's tails hold, and
the pixel goes to the redo pass with real work to do.  Timed per share: the default path (fast kernel + redo pass, caller's
workspace) and APGPU_STACK_SINGLE_KERNEL (the complete kernel alone), interleaved on one box.

    python tools/redo_sweep.py [--frames 64] [--size 4096] [--steps 10] > profiles/r05/redo_sweep.txt
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from astrophotography_amd import ops, synth  # noqa: E402


def timed(fn, steps):
    fn()
    torch.cuda.synchronize()
    evs = []
    for _ in range(steps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        fn()
        b.record()
        evs.append((a, b))
    torch.cuda.synchronize()
    t = sorted(a.elapsed_time(b) for a, b in evs)
    return t[len(t) // 2], t[0]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--frames', type=int, default=64)
    ap.add_argument('--size', type=int, default=4096)
    ap.add_argument('--steps', type=int, default=10)
    ap.add_argument('--rounds', type=int, default=2)
    ap.add_argument('--cases', default='', help='comma-separated substrings: run only the matching cases')
    args = ap.parse_args()
    dev = torch.device('cuda', 0)
    N, H, W = args.frames, args.size, args.size
    masters = synth.make_masters(H, W, config_id=2, device=dev)
    nflat, _ = ops.flat_normalize(masters['flat'])
    clean = synth.make_frames(N, masters, nflat, config_id=2, dtype=torch.float32, first_frame=0)
    calib = dict(bias=masters['bias'], dark=masters['dark'], nflat=nflat,
                 exp_ratio=torch.full((N,), synth.EXP_RATIO, dtype=torch.float32, device=dev))
    g = torch.Generator(device=dev)
    g.manual_seed(505)
    u = torch.rand(H * W, generator=g, device=dev).view(H, W)
    rows = torch.arange(H, device=dev).view(H, 1).expand(H, W)
    cases = [('natural', None), ('1 % random', u < 0.01), ('10 % random', u < 0.10), ('25 % random', u < 0.25), ('50 % random', u < 0.50),
             ('75 % random', u < 0.75),
             ('100 %', torch.ones_like(u, dtype=torch.bool)), ('top half (clustered)', rows < H // 2),
             ('bottom 10 % (clustered, late)', rows >= H - H // 10)]
    print('# %d x %d x %d float32, fused calibration, 3-sigma / maxiters 5 / median-centred clipped mean; median (min) of %d steps, %d interleaved rounds'
          % (N, H, W, args.steps, args.rounds))
    print('# %-32s %10s %22s %22s %8s %10s' % ('columns forced off the fast path', 'redo frac', 'fast + redo pass  ms', 'complete kernel  ms', 'ratio', 'first call'))
    frames = clean.clone()
    want = [c for c in args.cases.split(',') if c]
    for name, sel in cases:
        if want and not any(c in name for c in want):
            continue
        frames.copy_(clean)
        if sel is not None:
            for f in range(5):
                frames[7 * f + 3].add_(sel.to(torch.float32) * 3000.0)
        two, one = [], []
        # the first call on changed data runs in the mode the previous case left in the workspace (quiet = no guard): timed alone
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        ops.stack_sigclip(frames, sigma=3.0, maxiters=5, calib=calib, outputs=('mean',))
        b.record()
        torch.cuda.synchronize()
        first_ms = a.elapsed_time(b)
        for _ in range(args.rounds):
            ops.stack_redo_stats(reset=True)
            two.append(timed(lambda: ops.stack_sigclip(frames, sigma=3.0, maxiters=5, calib=calib, outputs=('mean',)), args.steps))
            st = ops.stack_redo_stats()
            one.append(timed(lambda: ops.stack_sigclip(frames, sigma=3.0, maxiters=5, calib=calib, outputs=('mean',), single_kernel=True), args.steps))
        t2 = min(t[0] for t in two)
        t1 = min(t[0] for t in one)
        print('%-34s %10.5f %12.4f (%7.4f) %12.4f (%7.4f) %8.3f %10.4f   listed %d + %d blocks per call'
              % (name, st['fraction'], t2, min(t[1] for t in two), t1, min(t[1] for t in one), t2 / t1, first_ms,
                 st['pixels_listed'] // max(st['calls'], 1), st['blocks_given_up'] // max(st['calls'], 1)))
        # same survivors either way
        a = ops.stack_sigclip(frames, sigma=3.0, maxiters=5, calib=calib, outputs=('mean', 'count'))
        b = ops.stack_sigclip(frames, sigma=3.0, maxiters=5, calib=calib, outputs=('mean', 'count'), single_kernel=True)
        assert torch.equal(a['count'], b['count']), name
        d = (a['mean'].view(torch.int32) - b['mean'].view(torch.int32)).abs().max().item()
        assert d <= 1, (name, d)


if __name__ == '__main__':
    main()
