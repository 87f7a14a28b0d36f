"""Development aid: 129 .. 512 frames with the median and std planes - chunk path + second pass (round 6) against the exact kernel.
   NS=256,512 python tools/bench_big_rich.py"""
import os
import sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from astrophotography_amd import ops, synth

H = W = 4096
dev = torch.device('cuda', 0)
masters = synth.make_masters(H, W, config_id=2, device=dev)
nflat, _ = ops.flat_normalize(masters['flat'])
for N in [int(x) for x in os.environ.get('NS', '512,384,300,257,256,192,129').split(',')]:
    frames = synth.make_frames(N, masters, nflat, config_id=2, dtype=torch.uint16 if os.environ.get('DT') == 'u16' else torch.float32, first_frame=0)
    calib = dict(bias=masters['bias'], dark=masters['dark'], nflat=nflat, exp_ratio=torch.full((N,), synth.EXP_RATIO, dtype=torch.float32, device=dev))

    def t(fn, reps=3):
        fn(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ts = []
        for _ in range(reps):
            e0.record(); out = fn(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
        return min(ts), out
    lean, _ = t(lambda: ops.stack_sigclip(frames, calib=calib, outputs=('mean', 'count')))
    rich, a = t(lambda: ops.stack_sigclip(frames, calib=calib, outputs=('mean', 'median', 'std', 'count')))
    exact, b = t(lambda: ops.stack_sigclip(frames, calib=calib, outputs=('mean', 'median', 'std', 'count'), exact=True), reps=1)
    same = bool(torch.equal(a['count'], b['count']))
    dm = (a['median'] - b['median']).abs().max().item()
    ds = ((a['std'] - b['std']).abs() / b['std'].clamp_min(1e-6)).max().item()
    gb = (N * frames.element_size() + 12 + 16) * H * W / 1e9
    print('N=%3d  mean+count %.3f ms | mean+median+std+count: chunk path %.3f ms (%.1f %% of 8 TB/s), exact kernel %.3f ms | counts equal %s, max |d median| %.3g, max rel d std %.3g' % (
        N, lean, rich, 100 * gb / (rich * 1e-3) / 8000, exact, same, dm, ds))
    del frames
