#!/usr/bin/env python3
"""ApFindBadPixels as one device pipeline (A3 -> thresholds -> A4), device time with the dark resident in HBM:
    python tools/bench_findbadpix.py [--size 4096]"""
import argparse
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from astrophotography_amd import ops, synth

ap = argparse.ArgumentParser()
ap.add_argument('--size', type=int, default=4096)
a = ap.parse_args()
dark = synth.make_masters(a.size, a.size, config_id=2, device='cuda')['dark']
sigma = 4.0


def pipeline():
    stats = ops.sigclip_global(dark, sigma=sigma, maxiters=5)
    thr = torch.stack((stats[1] - sigma * stats[2], stats[1] + sigma * stats[2]))
    return ops.threshold_mask(dark, thresholds=thr)


def t(fn, n=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


mask, nbad = pipeline()
print('ApFindBadPixels pipeline %dx%d float32, sigma 4, 5 iterations: %.3f ms device time (%d bad pixels)' % (
    a.size, a.size, t(pipeline), int(nbad.item())))
print('  sigclip_global alone   %.3f ms' % t(lambda: ops.sigclip_global(dark, sigma=sigma, maxiters=5)))
st = ops.sigclip_global(dark, sigma=sigma, maxiters=5)
thr = torch.stack((st[1] - sigma * st[2], st[1] + sigma * st[2]))
print('  threshold_mask alone   %.3f ms' % t(lambda: ops.threshold_mask(dark, thresholds=thr)))
m = synth.make_masters(a.size, a.size, config_id=2, device='cuda')
print('  flat_normalize         %.3f ms' % t(lambda: ops.flat_normalize(m['flat'])))
