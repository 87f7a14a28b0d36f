"""uint16 fused clipped stack, 64 x 4096^2 (development aid)."""
import os, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, 'tools'))
import torch
from astrophotography_amd import ops, synth
from bench_kernels import timeit
H = W = 4096
masters = synth.make_masters(H, W, config_id=2, device='cuda')
nflat, _ = ops.flat_normalize(masters['flat'])
f16 = synth.make_frames(64, masters, nflat, config_id=2, dtype=torch.uint16)
calib = dict(bias=masters['bias'], dark=masters['dark'], nflat=nflat, exp_ratio=synth.EXP_RATIO)
for _ in range(3):
    t, b = timeit(lambda: ops.stack_sigclip(f16, calib=calib, outputs=('mean',)), reps=7)
    print('u16 fused clipped mean 64x4096^2: %.3f ms (min %.3f)' % (t, b))
