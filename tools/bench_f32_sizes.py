#!/usr/bin/env python3
"""Fused clipped mean / median for EVERY stack size 2 .. 128 (development aid; profiles/<round>/bench_f32_sizes.txt and, with
--u16, bench_u16_sizes.txt).  A size is flagged when it costs more than the next FULL slot count - the 'holes' of the verdict."""
import os
import sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
sys.path.insert(0, os.path.join(root, 'tools'))
import torch
from astrophotography_amd import ops, synth
from bench_kernels import timeit

u16 = '--u16' in sys.argv
SLOTS = [1, 4, 8, 12, 16, 20, 24, 28, 32, 36, 40, 44, 48, 52, 56, 60, 64, 72, 80, 88, 96, 104, 112, 120, 128]
H = W = 4096
masters = synth.make_masters(H, W, config_id=2, device='cuda')
nflat, _ = ops.flat_normalize(masters['flat'])
frames = synth.make_frames(128, masters, nflat, config_id=2, dtype=torch.uint16 if u16 else torch.float32)
calib = dict(bias=masters['bias'], dark=masters['dark'], nflat=nflat, exp_ratio=synth.EXP_RATIO)
esz = 2 if u16 else 4
res = {}
for n in range(128, 1, -1):
    med, _ = timeit(lambda: ops.stack_sigclip(frames[:n], calib=calib, outputs=('mean',)), reps=5)
    med2, _ = timeit(lambda: ops.stack_median(frames[:n], calib=calib), reps=5)
    res[n] = (med, med2)
holes = 0
for n in range(128, 1, -1):
    med, med2 = res[n]
    nxt = min(s for s in SLOTS if s >= n)
    flag = ''
    if n != nxt and med > res[nxt][0] * 1.02:
        flag = '   <-- %.0f %% above N=%d' % (100 * (med / res[nxt][0] - 1), nxt)
        holes += 1
    print('N=%3d%s  clipped mean %.3f ms (%5.0f GB/s, %.1f us per frame)   median %.3f ms%s' % (
        n, ' *' if n == nxt else '  ', med, (esz * n + 16) * H * W / med / 1e6, 1e3 * med / n, med2, flag))
print('(* = a full slot count)   sizes costing > 2 %% more than the next full slot count: %d of %d' % (holes, 127 - len([s for s in SLOTS if s > 1])))
