#!/usr/bin/env python3
"""The ccdproc.combine configuration (A6: one pass of median / mad_std at 5 sigma, float64 mean + std + count) by frame count and
frame type: the fast kernel + rich kernel pair (stack_mad*.hip) against the rich kernel alone (APGPU_STACK_SINGLE_KERNEL), 4096 x 4096.

    python tools/bench_a6.py [--frames 16,32,64,96,128] > profiles/r05/bench_a6.txt
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from astrophotography_amd import ops  # noqa: E402


def timed(fn, n=6):
    fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        fn()
        b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    return sorted(ts)[len(ts) // 2]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--frames', default='16,32,64,96,128')
    ap.add_argument('--size', type=int, default=4096)
    a = ap.parse_args()
    kw = dict(sigma=5.0, maxiters=1, cenfunc='median', stdfunc='mad_std', outputs=('mean_f64', 'std_f64', 'count'))
    H = a.size
    print('# A6, %d x %d, float64 mean + std + count; median of 6 calls; roofline = (frame bytes + 20) per pixel at 8 TB/s' % (H, H))
    for N in [int(x) for x in a.frames.split(',')]:
        for dt in ('float32', 'uint16'):
            g = torch.Generator(device='cuda').manual_seed(3)
            fr = torch.randn((N, H, H), device='cuda', generator=g) * 12 + 1000
            if dt == 'uint16':
                fr = fr.round_().clamp_(0, 65535).to(torch.int32).to(torch.uint16)
            timed(lambda: ops.stack_sigclip(fr, **kw), 3)
            fast = timed(lambda: ops.stack_sigclip(fr, **kw))
            rich = timed(lambda: ops.stack_sigclip(fr, single_kernel=True, **kw), 3)
            x = ops.stack_sigclip(fr, **kw)
            y = ops.stack_sigclip(fr, single_kernel=True, **kw)
            same = bool(torch.equal(x['count'], y['count'])) and float((x['mean_f64'] - y['mean_f64']).abs().max()) == 0.0
            by = (N * fr.element_size() + 20) * H * H
            print('%3d frames %-7s  fast pair %7.3f ms (%4.1f %% of the roofline)   rich kernel alone %7.3f ms   identical results: %s'
                  % (N, dt, fast, 100 * by / fast / 1e6 / 8000, rich, same))
            del fr, x, y


if __name__ == '__main__':
    main()
