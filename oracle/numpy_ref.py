#!/usr/bin/env python3
"""Pure-NumPy restatement of the hot path, timed as the CPU baseline the north star names.
*** TEST INFRASTRUCTURE ONLY *** (tests/ and bench.py's cpu_baseline leg; never imported by the product).

  calibrate_numpy   = the expressions of ApCalibrate.calibrate (core/ApCalibrate.py:439-464), one temporary per
                      operation as the reference writes them, looped over the frames;
  sigclip_numpy     = the sigma clip along N the reference's own clip function performs,
                      astropy.stats.sigma_clipped_stats(cube, axis=0) in its NumPy form
                      (astropy/stats/sigma_clipping.py _sigmaclip_withaxis: nanmedian / nanstd along axis 0,
                      values outside the bounds -> NaN, repeat while anything changed, then nanmean) - checked
                      against oracle/apref.c (itself pinned by golden group G5) in tests/test_oracle_golden.py.

Run as a script it times `calibrate + clip` on a row sample, single process (NumPy elementwise operations and
nanmedian are single-threaded) and with multiprocessing over row blocks, and prints one JSON object:
    python oracle/numpy_ref.py --frames 64 --width 4096 --seconds 10 --workers 0
"""
import argparse
import json
import multiprocessing as mp
import os
import time
import warnings

import numpy as np


def calibrate_numpy(raw, bias, dark, nflat, exp_ratio, dark_still_biased=False):
    """core/ApCalibrate.py:439-464 for one frame (float32 in, float32 out; each line one NumPy temporary)."""
    if raw.dtype != np.float32:
        raw = raw.astype(np.float32)                        # ApCalibrate._read_fits :304-307
    img_sub_b = raw - bias                                  # :439
    dark_sub_b = dark - bias if dark_still_biased else dark  # :440-445
    dark_scaled = np.float32(exp_ratio) * dark_sub_b        # :450 (python float is a weak scalar against float32)
    img_sub_bd = img_sub_b - dark_scaled                    # :451
    if nflat is None:
        return img_sub_bd                                   # :472-474
    with np.errstate(divide='ignore', invalid='ignore'):
        return np.where(nflat != 0, img_sub_bd / nflat, img_sub_bd)     # :462-464


def sigclip_numpy(cube, sigma=3.0, maxiters=5):
    """sigma_clipped_stats(cube, axis=0) (median centre, std deviation): (mean float64 [..], count int32 [..])."""
    filtered = cube.astype(np.float64)                      # astropy works on a float64 copy
    filtered[~np.isfinite(filtered)] = np.nan
    it = 0
    with warnings.catch_warnings():
        warnings.simplefilter('ignore', RuntimeWarning)
        while maxiters is None or it < maxiters:
            it += 1
            cen = np.nanmedian(filtered, axis=0)
            std = np.nanstd(filtered, axis=0)
            lo, hi = cen - sigma * std, cen + sigma * std
            with np.errstate(invalid='ignore'):
                out = (filtered < lo) | (filtered > hi)
            if not out.any():
                break
            filtered[out] = np.nan
        mean = np.nanmean(filtered, axis=0)
    return mean, np.isfinite(filtered).sum(axis=0).astype(np.int32)


def calibrate_stack_numpy(raw, bias, dark, nflat, exp_ratio, sigma=3.0, maxiters=5):
    cal = np.empty(raw.shape, np.float32)
    for f in range(raw.shape[0]):
        cal[f] = calibrate_numpy(raw[f], bias, dark, nflat, exp_ratio)
    return sigclip_numpy(cal, sigma, maxiters)


def synth_block(n_frames, rows, width, seed):
    """A block of synthetic frames with the statistics of the bench workload (values only matter for the timing
    through the number of clip iterations: Gaussian noise + 0.1 % outliers)."""
    rng = np.random.default_rng(seed)
    shape = (rows, width)
    bias = rng.normal(1000, 5, shape).astype(np.float32)
    dark = rng.normal(20, 3, shape).astype(np.float32)
    nflat = rng.normal(1.0, 0.01, shape).astype(np.float32)
    raw = rng.normal(1600, 30, (n_frames,) + shape).astype(np.float32)
    hits = rng.random(raw.shape) < 0.001
    raw[hits] += 3000
    return raw, bias, dark, nflat


def _work(args):
    """Reduces synthetic blocks of `rows` rows until `seconds` have passed: (blocks done, compute seconds)."""
    n_frames, rows, width, seed, seconds = args
    raw, bias, dark, nflat = synth_block(n_frames, rows, width, seed)
    t0 = time.perf_counter()
    n = 0
    while True:
        calibrate_stack_numpy(raw, bias, dark, nflat, 0.4)
        n += 1
        if time.perf_counter() - t0 >= seconds:
            break
    return n, time.perf_counter() - t0


def cpu_model():
    try:
        for ln in open('/proc/cpuinfo'):
            if ln.startswith('model name'):
                return ln.split(':', 1)[1].strip()
    except OSError:
        pass
    return 'unknown CPU'


def time_numpy(n_frames=64, width=4096, seconds=10.0, workers=0):
    """Mpixels/s (input frame pixels) of the NumPy path: single process, and `workers` processes (0 = os.cpu_count())
    each reducing its own row blocks.  Bounded: every leg stops after `seconds` (+ at most one block)."""
    rows = 8
    _work((n_frames, 2, width, 0, 0.0))                     # warm
    n1, t1 = _work((n_frames, rows, width, 2, seconds))
    single = n1 * n_frames * rows * width / 1e6 / t1
    workers = workers or os.cpu_count() or 1
    res = dict(single=dict(value=single, unit='Mpixels/s', cores=1, rows_per_block=rows, blocks=n1, seconds=t1), cpu=cpu_model(),
               numpy=np.__version__)
    if workers > 1:
        rows_w = 2                                          # small blocks: with every core busy a block takes much longer
        ctx = mp.get_context('fork')
        t0 = time.perf_counter()
        with ctx.Pool(workers) as pool:
            out = pool.map(_work, [(n_frames, rows_w, width, 10 + w, seconds) for w in range(workers)], chunksize=1)
        wall = time.perf_counter() - t0
        # aggregate throughput of the compute phase: all blocks done over the slowest worker's compute time
        blocks = sum(n for n, _ in out)
        tmax = max(t for _, t in out)
        res['multi'] = dict(value=blocks * n_frames * rows_w * width / 1e6 / tmax, unit='Mpixels/s', cores=workers,
                            rows_per_block=rows_w, blocks=blocks, seconds=tmax, wall_seconds_incl_synthesis=wall)
    return res


if __name__ == '__main__':
    ap = argparse.ArgumentParser()
    ap.add_argument('--frames', type=int, default=64)
    ap.add_argument('--width', type=int, default=4096)
    ap.add_argument('--seconds', type=float, default=10.0)
    ap.add_argument('--workers', type=int, default=0)
    a = ap.parse_args()
    print(json.dumps(time_numpy(a.frames, a.width, a.seconds, a.workers)))
