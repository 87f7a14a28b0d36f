#!/usr/bin/env python3
"""Re-runs one case of tools/fuzz_long.py and prints the mismatching columns (development aid)."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from astrophotography_amd import ops
from oracle import apref
from tests.util import synth_cube, synth_masters
from tools.fuzz_long import dev

np.set_printoptions(precision=9, linewidth=200)
for seed in [int(s) for s in sys.argv[1:]]:
    rng = np.random.default_rng(seed)
    N = int(rng.choice([int(rng.integers(1, 129)), int(rng.integers(129, 513)), int(rng.choice([40, 56, 72, 80, 100, 112, 120, 128, 129, 256, 512]))]))
    H, W = int(rng.integers(1, 6)), int(rng.integers(1, 200))
    u16 = bool(rng.integers(0, 2))
    cube = synth_cube(rng, N, (H, W), nan_frac=0.0 if u16 else float(rng.choice([0.0, 0.02, 0.3])), dtype=np.uint16 if u16 else np.float32)
    bias, dark, flat = synth_masters(rng, (H, W))
    nflat = (flat / np.float32(30000.0)).astype(np.float32)
    if rng.integers(0, 2) and W > 6:
        nflat[0, :6] = [0.0, -1.5, np.nan, 1e-30, 1e30, np.inf]
    if rng.integers(0, 3) == 0:
        nflat = None
    e = np.full(N, 0.4, np.float32) if rng.integers(0, 2) else rng.uniform(0.2, 0.6, N).astype(np.float32)
    ped = np.where(rng.random(N) < 0.3, -50.0, 0.0) if rng.integers(0, 3) == 0 else None
    sb = bool(rng.integers(0, 2))
    use_calib = bool(rng.integers(0, 4) > 0)
    if use_calib:
        cal = apref.calibrate(cube, bias, dark, nflat, e, ped, dark_still_biased=sb)
        calib = dict(bias=dev(bias), dark=dev(dark), nflat=None if nflat is None else dev(nflat), exp_ratio=dev(e), pedestal=ped, dark_still_biased=sb)
    else:
        cal, calib = cube.astype(np.float32), None
    pixmask = (rng.random((H, W)) < 0.05).astype(np.uint8) if rng.integers(0, 2) else None
    sigma = float(rng.choice([0.5, 1.5, 2.0, 3.0, 5.0]))
    mi = rng.choice([1, 2, 5, None])
    mi = None if mi is None else int(mi)
    cen = str(rng.choice(['median', 'mean']))
    dv = str(rng.choice(['std', 'std', 'mad_std']))
    outs_i = int(rng.integers(0, 7))
    sl, su = (sigma, sigma) if rng.integers(0, 3) else (float(rng.choice([0.5, 1.25, 2.0, 3.0, 1e30])), float(rng.choice([0.5, 1.5, 3.0, 4.0, 1e30])))
    with np.errstate(all='ignore'):
        ref = apref.stack_sigclip(cal, sigma_lower=sl, sigma_upper=su, maxiters=mi, cenfunc=cen, stdfunc=dv, pixmask=pixmask)
    r = ops.stack_sigclip(dev(cube), sigma_lower=sl, sigma_upper=su, maxiters=mi, cenfunc=cen, stdfunc=dv, calib=calib,
                          pixmask=None if pixmask is None else dev(pixmask), outputs=('mean', 'count'))
    got = r['count'].cpu().numpy()
    gm = r['mean'].cpu().numpy()
    bad = np.argwhere((got != ref['count']) | ~((gm == ref['mean'].astype(np.float32)) | (np.isnan(gm) & np.isnan(ref['mean']))))
    sigma = (sl, su)
    print(f'seed {seed}: N={N} {H}x{W} u16={u16} calib={use_calib} {cen}/{dv} sigma={sigma} maxiters={mi}: {len(bad)} of {H * W} columns differ')
    for (y, x) in bad[:3]:
        col = np.sort(cal[:, y, x].astype(np.float64))
        print('  pixel', y, x, 'oracle count', ref['count'][y, x], 'gpu count', got[y, x], 'oracle lo/hi', ref['lo'][y, x], ref['hi'][y, x],
              'oracle mean', ref['mean'][y, x], 'gpu mean', float(r['mean'][y, x]))
        print('  sorted column:', col[:12], '...' if len(col) > 12 else '', 'n finite', np.isfinite(col).sum())
        keep = ref['keep'][:, y, x]
        print('  oracle survivors:', np.sort(cal[keep, y, x]))
