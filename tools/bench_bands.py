"""Development aid (round 6): C5's share in ROW BANDS - resample a band of output rows into a scratch buffer that is reused band
after band, clip it at once - so that the resampled values are read back from the 256 MB Infinity Cache instead of HBM.
Timing experiment: the band's origin is folded into the affine coefficients (NOT the bit-exact definition; results unchecked).
   python tools/bench_bands.py [N H W]"""
import sys
import numpy as np
import torch
sys.path.insert(0, '.')
from astrophotography_amd import ops, synth

args = [a for a in sys.argv[1:] if not a.startswith('--')]
N, H, W = (int(args[0]), int(args[1]), int(args[2])) if len(args) >= 3 else (16, 8192, 8192)
dev = torch.device('cuda', 0)
masters = synth.make_masters(H, W, config_id=2, device=dev)
nflat, _ = ops.flat_normalize(masters['flat'])
frames = synth.make_frames(N, masters, nflat, config_id=2, dtype=torch.float32, first_frame=0)
cal = ops.calibrate(frames, masters['bias'], masters['dark'], nflat, synth.EXP_RATIO)
del frames
st = ops.sigclip_global(masters['dark'], sigma=4.0, maxiters=5)
badmask, _ = ops.threshold_mask(masters['dark'], thresholds=st[3:5].contiguous())
rng = np.random.default_rng(5000)
th = np.deg2rad(rng.uniform(-0.2, 0.2, N))
A = np.stack([np.cos(th), -np.sin(th), rng.uniform(-3, 3, N), np.sin(th), np.cos(th), rng.uniform(-3, 3, N)], 1)
out = torch.empty((H, W), dtype=torch.float32, device=dev)


def run(rows, use_mask):
    scratch = torch.empty((N, rows, W), dtype=torch.float32, device=dev)
    affs = []
    for y0 in range(0, H, rows):
        B = A.copy()
        B[:, 2] += A[:, 1] * y0
        B[:, 5] += A[:, 4] * y0
        affs.append(torch.as_tensor(B, dtype=torch.float64))

    def step():
        for i, y0 in enumerate(range(0, H, rows)):
            ops.resample_affine(cal, affs[i], mask=badmask if use_mask else None, out=scratch, out_shape=(rows, W), weight=False)
            out[y0:y0 + rows] = ops.stack_sigclip(scratch, sigma=3.0, maxiters=5, outputs=('mean',))['mean']
    for _ in range(2):
        step()
    torch.cuda.synchronize()
    ts = []
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(6):
        e0.record()
        step()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    return np.median(ts), min(ts)


for use_mask in (False, True):
    for rows in (8192, 2048, 1024, 512, 256, 128):
        if rows > H:
            continue
        m, b = run(rows, use_mask)
        print('mask=%s band of %5d rows (%6.1f MB of resampled values): median %.3f ms  min %.3f ms' % (use_mask, rows, N * rows * W * 4 / 1e6, m, b))
