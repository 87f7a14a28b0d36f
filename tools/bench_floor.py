#!/usr/bin/env python3
"""Development aid: the fused 64-frame kernel on slabs small enough for the Infinity Cache (no HBM traffic on repeated
launches) against the full-size slab - per-pixel time; says how much of the full-size time is memory, not instructions."""
import os
import sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
sys.path.insert(0, os.path.join(root, 'tools'))
import torch
from astrophotography_amd import ops, synth
from bench_kernels import timeit

N = int(sys.argv[1]) if len(sys.argv) > 1 else 64
for H in (512, 768, 1024, 1536, 2048, 4096):
    W = H
    masters = synth.make_masters(H, W, config_id=2, device='cuda')
    nflat, _ = ops.flat_normalize(masters['flat'])
    frames = synth.make_frames(N, masters, nflat, config_id=2)
    calib = dict(bias=masters['bias'], dark=masters['dark'], nflat=nflat, exp_ratio=synth.EXP_RATIO)
    med, mn = timeit(lambda: ops.stack_sigclip(frames, calib=calib, outputs=('mean',)), reps=9, batch=8 if H <= 1024 else 2)
    px = H * W
    print('%4d^2 (%6.0f MB slab): %.4f ms  min %.4f   %.3f ns per 1000 pixels   %5.0f GB/s algorithmic' % (
        H, 4e-6 * N * px, med, mn, 1e9 * med / px, (4 * N + 16) * px / med / 1e6))
    del frames, masters, nflat
    torch.cuda.empty_cache()
