#!/usr/bin/env python3
"""Per-kernel throughput of the whole path on one MI355X (development/evidence aid, not the contract
bench): algorithmic GB/s of every entry point of libapgpu.so against the 8 TB/s HBM peak."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from astrophotography_amd import ops, synth


def timeit(fn, reps=10, warm=2, batch=1):
    """Median / min device time per call.  batch > 1 queues that many calls between the two events so
    that sub-100-us kernels are not timed through the host's launch gaps."""
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(batch):
            fn()
        b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) / batch)
    ts.sort()
    return ts[len(ts) // 2], ts[0]


def main():
    H = W = 4096
    P = H * W
    N = 64
    masters = synth.make_masters(H, W, config_id=2, device='cuda')
    nflat, _ = ops.flat_normalize(masters['flat'])
    frames = synth.make_frames(N, masters, nflat, config_id=2)
    calib = dict(bias=masters['bias'], dark=masters['dark'], nflat=nflat, exp_ratio=synth.EXP_RATIO)
    rows = []

    only = sys.argv[1] if len(sys.argv) > 1 else ''

    def rec(name, bytes_, fn, **kw):
        if only and only not in name:
            return
        med, best = timeit(fn, **kw)
        rows.append(dict(kernel=name, ms=med, ms_min=best, GBps=bytes_ / med / 1e6, frac_of_8TBps=bytes_ / med / 1e6 / 8000))
        print('%-46s %8.3f ms  %7.0f GB/s  %5.1f %%' % (name, med, bytes_ / med / 1e6, 100 * bytes_ / med / 1e6 / 8000), flush=True)

    out = torch.empty_like(frames)
    rec('calibrate f32 slab 64x4096^2 (A2)', (4 + 4) * N * P + 12 * P, lambda: ops.calibrate(frames, masters['bias'], masters['dark'], nflat, synth.EXP_RATIO, out=out))
    del out
    rec('stack_sigclip fused, mean (A2+A7, bench kernel)', 4 * N * P + 16 * P, lambda: ops.stack_sigclip(frames, calib=calib, outputs=('mean',)))
    rec('stack_sigclip fused, moments (N-shard partials)', 4 * N * P + 24 * P, lambda: ops.stack_sigclip(frames, calib=calib, outputs=('moments',)))
    rec('stack_sigclip fused, mean+median+std (EXTRA)', 4 * N * P + 24 * P, lambda: ops.stack_sigclip(frames, calib=calib, outputs=('mean', 'median', 'std')), reps=5)
    rec('stack_sigclip plain f32 (A7, no calibration)', 4 * N * P + 4 * P, lambda: ops.stack_sigclip(frames, outputs=('mean',)))
    rec('stack ccdproc config: 1 pass, median/mad_std 5s (A6)', 4 * N * P + 4 * P, lambda: ops.stack_sigclip(frames, sigma=5.0, maxiters=1, stdfunc='mad_std', outputs=('mean',)), reps=3, warm=1)
    rec('stack_sigclip fused, 48 of 64 slots (non-FULL lean)', 4 * 48 * P + 16 * P, lambda: ops.stack_sigclip(frames[:48], calib=calib, outputs=('mean',)))
    rec('stack_median fused (C4-style, f32)', 4 * N * P + 16 * P, lambda: ops.stack_median(frames, calib=calib))
    f16 = synth.make_frames(N, masters, nflat, config_id=2, dtype=torch.uint16)
    rec('stack_sigclip fused u16 raw', 2 * N * P + 16 * P, lambda: ops.stack_sigclip(f16, calib=calib, outputs=('mean',)))
    rec('stack_median fused u16 raw (C4-style)', 2 * N * P + 16 * P, lambda: ops.stack_median(f16, calib=calib))
    del f16
    img = frames[0].contiguous()
    rec('flat_normalize 4096^2 (A1)', 12 * P, lambda: ops.flat_normalize(masters['flat']), batch=20)
    rec('sigclip_global f32 4096^2, sigma 4, 5 iters (A3)', 4 * P * 10 * 3, lambda: ops.sigclip_global(masters['dark'], sigma=4.0, maxiters=5), reps=5)
    rec('threshold_mask 4096^2 (A4)', 5 * P, lambda: ops.threshold_mask(masters['dark'], 5.0, 35.0), batch=20)
    mask, _ = ops.threshold_mask(masters['dark'], 5.0, 35.0)
    rec('fix_badpix 4096^2, delta 2 (A5)', 9 * P, lambda: ops.fix_badpix(img, mask, 2), batch=20)
    print('bad pixel fraction %.4f' % float(mask.float().mean()))
    rec('imarith f32 SUB image 4096^2 (A8)', 12 * P, lambda: ops.imarith(img, 'SUB', masters['bias']), batch=20)
    import numpy as np
    rng = np.random.default_rng(5)
    th = np.deg2rad(rng.uniform(-0.2, 0.2, N))
    A = np.stack([np.cos(th), -np.sin(th), rng.uniform(-3, 3, N), np.sin(th), np.cos(th), rng.uniform(-3, 3, N)], 1)
    out = torch.empty_like(frames)
    rec('resample_affine 64x4096^2 f32, Lanczos-3 (F3)', 8 * N * P, lambda: ops.resample_affine(frames, A, out=out, weight=False), reps=5)
    rec('resample_affine + uint8 weight planes', 9 * N * P, lambda: ops.resample_affine(frames, A, out=out, weight=True), reps=5)
    del out
    json.dump(rows, open(os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'gpurun_out', 'bench_kernels.json'), 'w'), indent=1)


if __name__ == '__main__':
    main()
