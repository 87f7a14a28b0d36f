"""Development aid: 129..512-frame stacks - chunked fast kernel vs the exact LDS kernel, timing and agreement."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from astrophotography_amd import ops, synth

H = int(os.environ.get('H', 4096)); W = 4096
dev = torch.device('cuda', 0)
masters = synth.make_masters(H, W, config_id=3, device=dev)
nflat, _ = ops.flat_normalize(masters['flat'])


def t(fn, n=5):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n


for N in [int(x) for x in os.environ.get('NS', '256,192,200,129').split(',')]:
    frames = synth.make_frames(N, masters, nflat, config_id=3)
    calib = dict(bias=masters['bias'], dark=masters['dark'], nflat=nflat, exp_ratio=torch.full((N,), synth.EXP_RATIO, device=dev), dark_still_biased=False)
    ops.stack_redo_stats(reset=True)
    fast = ops.stack_sigclip(frames, calib=calib, outputs=('mean', 'count'))
    st = ops.stack_redo_stats()
    exact = ops.stack_sigclip(frames, calib=calib, outputs=('mean', 'count'), exact=True)
    torch.cuda.synchronize()
    same_cnt = bool(torch.equal(fast['count'], exact['count']))
    d = (fast['mean'].view(torch.int32).long() - exact['mean'].view(torch.int32).long()).abs()
    print('N=%d %s: redone exactly %.4f of the pixels, counts equal %s, mean max ulp %d, exact-equal %.4f | fast %.3f ms, exact %.3f ms' % (
        N, ops.stack_kernel_name(N, 'f32', True), st['fraction'], same_cnt, int(d.max()), float((d == 0).double().mean()),
        t(lambda: ops.stack_sigclip(frames, calib=calib, outputs=('mean',))), t(lambda: ops.stack_sigclip(frames, calib=calib, outputs=('mean',), exact=True), 2)))
    del frames
