"""Times the resample kernels alone (development aid): python tools/bench_resample.py [--frames 64] [--size 4096] [--rot 0.2]
[--os 1] [--reps 5] [--two-step].  Prints ms per call and the fraction of the 8 TB/s HBM peak for 8 bytes per OUTPUT pixel."""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from astrophotography_amd import ops  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--frames', type=int, default=64)
    ap.add_argument('--size', type=int, default=4096)
    ap.add_argument('--rot', type=float, default=0.2, help='rotations drawn from +-rot degrees')
    ap.add_argument('--os', type=int, default=1)
    ap.add_argument('--scale', type=float, default=1.0, help='input pixels per output pixel')
    ap.add_argument('--reps', type=int, default=5)
    ap.add_argument('--two-step', action='store_true')
    ap.add_argument('--weight', action='store_true')
    a = ap.parse_args()
    N, H = a.frames, a.size
    g = torch.Generator(device='cuda').manual_seed(1)
    frames = torch.rand((N, H, H), device='cuda', generator=g) * 1000.0
    rng = np.random.default_rng(5)
    th = np.deg2rad(rng.uniform(-a.rot, a.rot, N))
    sc = a.scale
    A = np.stack([sc * np.cos(th), -sc * np.sin(th), rng.uniform(-3, 3, N), sc * np.sin(th), sc * np.cos(th), rng.uniform(-3, 3, N)], 1)
    if a.os > 1:
        fn = (ops.resample_oversampled_two_step if a.two_step else ops.resample_oversampled)
        call = lambda: fn(frames, A, a.os)
    else:
        out = torch.empty_like(frames)
        call = lambda: ops.resample_affine(frames, A, out=out, weight=a.weight)
    call()
    torch.cuda.synchronize()
    ts = []
    for _ in range(a.reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        call()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    ms = float(np.median(ts))
    by = 8.0 * N * H * H
    print('resample %dx%d^2 os=%d%s: %.3f ms (min %.3f)  %.0f GB/s  %.1f %% of 8 TB/s' %
          (N, H, a.os, ' two-step' if a.two_step else '', ms, min(ts), by / ms / 1e6, 100 * by / ms / 1e6 / 8000))


if __name__ == '__main__':
    main()
