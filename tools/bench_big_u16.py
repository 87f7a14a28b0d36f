import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from astrophotography_amd import ops, synth
dev = torch.device('cuda', 0)
masters = synth.make_masters(4096, 4096, config_id=3, device=dev)
nflat, _ = ops.flat_normalize(masters['flat'])
for N in (129, 200, 257, 300, 448, 256, 512):
    frames = synth.make_frames(N, masters, nflat, config_id=3, dtype=torch.uint16)
    calib = dict(bias=masters['bias'], dark=masters['dark'], nflat=nflat, exp_ratio=torch.full((N,), synth.EXP_RATIO, device=dev))
    for cal in (calib, None):
        fn = lambda: ops.stack_sigclip(frames, calib=cal, outputs=('mean', 'count'))
        fn(); torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(5): fn()
        b.record(); torch.cuda.synchronize()
        print('u16 lean N=%d calibrated=%s %.3f ms' % (N, cal is not None, a.elapsed_time(b) / 5))
    del frames
