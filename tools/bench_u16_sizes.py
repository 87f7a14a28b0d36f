import sys, os
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import torch
from astrophotography_amd import ops, synth
sys.path.insert(0, os.path.join(os.environ.get('GRAFT_REPO_ROOT', '/root/repo'), 'tools'))
from bench_kernels import timeit
H=W=4096; P=H*W
masters = synth.make_masters(H, W, config_id=2, device='cuda')
nflat,_ = ops.flat_normalize(masters['flat'])
f16 = synth.make_frames(128, masters, nflat, config_id=2, dtype=torch.uint16)
calib = dict(bias=masters['bias'], dark=masters['dark'], nflat=nflat, exp_ratio=synth.EXP_RATIO)
for n in (128, 112, 100, 96, 80, 64, 56, 48, 40, 37, 32, 24, 20, 16, 12, 8):
    med, best = timeit(lambda: ops.stack_sigclip(f16[:n], calib=calib, outputs=('mean',)))
    print('u16 sigclip N=%d: %.3f ms' % (n, med))
    med, best = timeit(lambda: ops.stack_median(f16[:n], calib=calib))
    print('u16 median  N=%d: %.3f ms' % (n, med))
