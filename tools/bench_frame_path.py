#!/usr/bin/env python3
"""Per-frame path of the reference's calibrate_all.sh on one MI355X (development / evidence aid, not the contract bench):
raw frame -> bias / dark / flat calibration -> bad-pixel repair -> L.A.Cosmic -> sky-background mesh -> background subtraction,
the frame and the masters resident in HBM.  Beside it, with --cpu, the oracle restatements (numpy / scipy, one process) of the
two F4 steps on a bounded sample (a 1024^2 crop), scaled to the frame by pixel count.

    python tools/bench_frame_path.py [--size 4096] [--cpu]
"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from astrophotography_amd import ops, synth
from astrophotography_amd.core.ApFixCosmicRays import ApFixCosmicRays
from astrophotography_amd.core.ApMeasureBackground import ApMeasureBackground


def wall(fn, reps=5, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        ts.append(1e3 * (time.perf_counter() - t0))
    return sorted(ts)[len(ts) // 2]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--size', type=int, default=4096)
    ap.add_argument('--cpu', action='store_true')
    a = ap.parse_args()
    H = W = a.size
    masters = synth.make_masters(H, W, config_id=2, device='cuda')
    nflat, _ = ops.flat_normalize(masters['flat'])
    sky, _ = synth.make_sky_frame(H, W)
    # a raw frame whose calibration gives back the sky frame (plus the synthetic hot pixels of the dark)
    raw = (masters['bias'] + synth.EXP_RATIO * masters['dark'] + nflat * sky).clamp(0, 65535).round().to(torch.float32).contiguous()
    badmask, _ = ops.threshold_mask(masters['dark'], 5.0, 35.0)
    crfix, bkg = ApFixCosmicRays('ERROR'), ApMeasureBackground('ERROR')
    out = {}

    def calibrate():
        return ops.calibrate(raw, masters['bias'], masters['dark'], nflat, synth.EXP_RATIO)

    cal = calibrate()
    fixed, _ = ops.fix_badpix(cal, badmask, 2)
    clean, crmask, _ = crfix.process_tensor(fixed, 1.0)

    def whole():
        c = calibrate()
        f, _ = ops.fix_badpix(c, badmask, 2)
        cl, _, _ = crfix.process_tensor(f, 1.0)
        bkg.process_data(cl)
        return ops.imarith(cl.double(), 'SUB', bkg.get_bgimage_device())

    steps = [('calibrate (bias, dark, flat)', calibrate), ('fix_badpix delta 2', lambda: ops.fix_badpix(cal, badmask, 2)),
             ('ApFixCosmicRays.process_tensor', lambda: crfix.process_tensor(fixed, 1.0)),
             ('ApMeasureBackground.process_data', lambda: bkg.process_data(clean)),
             ('whole per-frame path', whole)]
    for name, fn in steps:
        out[name] = wall(fn)
        print('%-40s %9.3f ms' % (name, out[name]), flush=True)
    print('cosmic-ray pixels flagged: %d' % int(crmask.sum()))
    print('frames per second through the whole path: %.1f (%d x %d)' % (1e3 / out['whole per-frame path'], H, W))
    if a.cpu:
        from oracle import background_ref as br
        from oracle import lacosmic_ref as lr
        n = 1024
        crop = clean[:n, :n].cpu().numpy()
        t0 = time.perf_counter()
        lr.detect_cosmics(fixed[:n, :n].cpu().numpy(), gain=1.0, satlevel=65535.0)
        t_cr = time.perf_counter() - t0
        t0 = time.perf_counter()
        m = br.make_source_mask(crop)[0]
        br.background2d(crop, m, 66, 66)
        t_bg = time.perf_counter() - t0
        scale = (H * W) / float(n * n)
        out['cpu_numpy_lacosmic_ms_scaled'] = 1e3 * t_cr * scale
        out['cpu_numpy_background_ms_scaled'] = 1e3 * t_bg * scale
        print('CPU (oracle restatement, numpy/scipy, 1 process, %d^2 crop scaled by %.0f): L.A.Cosmic %.0f ms, background %.0f ms per frame'
              % (n, scale, out['cpu_numpy_lacosmic_ms_scaled'], out['cpu_numpy_background_ms_scaled']))
    json.dump(out, open(os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'gpurun_out', 'bench_frame_path.json'), 'w'), indent=1)


if __name__ == '__main__':
    main()
