import sys, os
root = os.environ.get('GRAFT_REPO_ROOT', '/root/repo')
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, 'tools'))
import torch
from astrophotography_amd import ops, synth
from bench_kernels import timeit
H=W=4096
masters = synth.make_masters(H, W, config_id=2, device='cuda')
nflat,_ = ops.flat_normalize(masters['flat'])
frames = synth.make_frames(64, masters, nflat, config_id=2)
f16 = synth.make_frames(64, masters, nflat, config_id=2, dtype=torch.uint16)
for name, fn in (('plain f32 median', lambda: ops.stack_median(frames)), ('plain u16 median', lambda: ops.stack_median(f16)),
                 ('plain f32 sigclip', lambda: ops.stack_sigclip(frames, outputs=('mean',))), ('plain u16 sigclip', lambda: ops.stack_sigclip(f16, outputs=('mean',)))):
    med, best = timeit(fn)
    print('%-20s %.3f ms' % (name, med))
