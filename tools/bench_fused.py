"""Development aid: C5's per-GPU share (16 x 8192^2, bad-pixel mask, +-0.2 degrees, +-3 pixels) - the two-step form
(resample_affine + stack_sigclip) against the fused launch (resample_stack_sigclip), same inputs, HIP-event times.
   python tools/bench_fused.py [N H W] [--nomask]"""
import sys
import numpy as np
import torch
sys.path.insert(0, '.')
from astrophotography_amd import ops, synth

args = [a for a in sys.argv[1:] if not a.startswith('--')]
N, H, W = (int(args[0]), int(args[1]), int(args[2])) if len(args) >= 3 else (16, 8192, 8192)
use_mask = '--nomask' not in sys.argv
dev = torch.device('cuda', 0)
masters = synth.make_masters(H, W, config_id=2, device=dev)
nflat, _ = ops.flat_normalize(masters['flat'])
frames = synth.make_frames(N, masters, nflat, config_id=2, dtype=torch.float32, first_frame=0)
cal = ops.calibrate(frames, masters['bias'], masters['dark'], nflat, synth.EXP_RATIO)
del frames
st = ops.sigclip_global(masters['dark'], sigma=4.0, maxiters=5)
badmask, _ = ops.threshold_mask(masters['dark'], thresholds=st[3:5].contiguous())
if not use_mask:
    badmask = None
rng = np.random.default_rng(5000)
th = np.deg2rad(rng.uniform(-0.2, 0.2, N))
A = np.stack([np.cos(th), -np.sin(th), rng.uniform(-3, 3, N), np.sin(th), np.cos(th), rng.uniform(-3, 3, N)], 1)
resampled = torch.empty_like(cal)


def two_step():
    ops.resample_affine(cal, A, mask=badmask, out=resampled, weight=False)
    return ops.stack_sigclip(resampled, sigma=3.0, maxiters=5, outputs=('mean',))['mean']


def fused():
    return ops.resample_stack_sigclip(cal, A, mask=badmask, sigma=3.0, maxiters=5, outputs=('mean',))['mean']


def timeit(fn, steps=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ts = []
    for _ in range(steps):
        e0.record()
        out = fn()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    return np.median(ts), min(ts), out


for rep in range(1 if "--fusedonly" in sys.argv else 2):
    m2, b2, o2 = timeit(two_step) if '--fusedonly' not in sys.argv else (0.0, 0.0, None)
    mf, bf, of = timeit(fused)
    print('N=%d %dx%d mask=%s  two-step: median %.3f min %.3f ms   fused: median %.3f min %.3f ms' % (N, H, W, use_mask, m2, b2, mf, bf))
if o2 is None:
    sys.exit(0)
same = torch.equal(torch.nan_to_num(o2, nan=-1.0), torch.nan_to_num(of, nan=-1.0))
d = (o2 - of).abs()
print('outputs bit-equal:', same, ' max |diff| %.3g  NaN positions equal: %s' % (float(torch.nan_to_num(d, nan=0.0).max()), bool(torch.equal(torch.isnan(o2), torch.isnan(of)))))
