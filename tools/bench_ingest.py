#!/usr/bin/env python3
"""Ingest -> stack -> egress of a set of FITS files, timed step by step (profiles/<round>/bench_ingest.txt):
    python tools/bench_ingest.py [--frames 64] [--height 4096] [--width 4096] [--dir /tmp/apingest]
Writes N uint16 FITS files (BZERO 32768, as cameras do), then times
  read_slab_device   disk -> pinned staging (reader thread) | H2D + on-device decode into the slab (copy stream), overlapped
  host path          the round-2 route for comparison: fitsio.read per file -> np.stack -> pageable upload
  stack              ccdproc-configuration clipped mean of the slab (ApMasterCal's kernel call)
  write              float64 master -> FITS (device encode + file write)
The files sit in the page cache after being written, so 'read' measures the copy out of the cache, not the disk."""
import argparse
import os
import shutil
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from astrophotography_amd import fitsio, ops


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--frames', type=int, default=64)
    ap.add_argument('--height', type=int, default=4096)
    ap.add_argument('--width', type=int, default=4096)
    ap.add_argument('--dir', default='/tmp/apingest')
    a = ap.parse_args()
    os.makedirs(a.dir, exist_ok=True)
    rng = np.random.default_rng(1)
    files = []
    base = rng.normal(1000, 30, (a.height, a.width))
    t0 = time.perf_counter()
    for i in range(a.frames):
        fr = np.clip(np.rint(base + rng.normal(0, 5, (1, a.width))), 0, 65535).astype(np.uint16)
        p = os.path.join(a.dir, 'f%03d.fits' % i)
        fitsio.write(p, fr, None, overwrite=True)
        files.append(p)
    gb = a.frames * a.height * a.width * 2 / 1e9
    print('wrote %d files (%.2f GB) in %.1f s' % (a.frames, gb, time.perf_counter() - t0))
    torch.cuda.init()
    torch.zeros(1, device='cuda')

    def sync_time(fn):
        torch.cuda.synchronize()
        t = time.perf_counter()
        r = fn()
        torch.cuda.synchronize()
        return r, time.perf_counter() - t

    fitsio.read_slab_device(files[:2])                      # warm: library load, pinned allocator
    tm = {}
    (slab, hdrs), t_ing = sync_time(lambda: fitsio.read_slab_device(files, timings=tm))
    print('read_slab_device: %.3f s total (%.2f GB/s)  | headers %.3f s, disk->pinned %.3f s (reader thread), H2D + decode overlapped' % (
        t_ing, gb / t_ing, tm['headers'], tm['read']))

    def host_path():
        arrs = [fitsio.read(f)[0] for f in files]
        return ops.to_device_u16(np.stack(arrs, 0))
    _, t_host = sync_time(host_path)
    print('host path (read + np.stack + pageable upload): %.3f s (%.2f GB/s)' % (t_host, gb / t_host))
    ops.stack_sigclip(slab[:, :64], sigma=5.0, maxiters=1, cenfunc='median', stdfunc='mad_std', outputs=('mean_f64', 'count', 'std_f64'))
    res, t_stack = sync_time(lambda: ops.stack_sigclip(slab, sigma=5.0, maxiters=1, cenfunc='median', stdfunc='mad_std',
                                                        outputs=('mean_f64', 'count', 'std_f64')))
    print('stack (ccdproc configuration, float64 mean / std / count): %.4f s' % t_stack)
    _, t_lean = sync_time(lambda: ops.stack_sigclip(slab, sigma=3.0, maxiters=5, outputs=('mean',)))
    print('stack (3-sigma clipped mean, lean kernel): %.4f s' % t_lean)
    _, t_wr = sync_time(lambda: fitsio.write_device(os.path.join(a.dir, 'master.fits'), res['mean_f64'], hdrs[0], overwrite=True))
    print('write master (float64, device encode + file): %.3f s' % t_wr)
    print('end to end (device ingest + stack + write): %.3f s; with the host path: %.3f s' % (t_ing + t_stack + t_wr, t_host + t_stack + t_wr))
    shutil.rmtree(a.dir, ignore_errors=True)


if __name__ == '__main__':
    main()
