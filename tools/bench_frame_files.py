#!/usr/bin/env python3
"""End-to-end per-frame cost of the reference's real flow ON FILES: scripts/calibrate_all.sh:406-411 starts
one `ap_calibrate.py` process per raw frame (bias / dark / flat / bad-pixel repair; here without --fixcosmic and with it).
Measured on one MI355X box, files on the box's local disk (/tmp):
  (a) one process per frame, exactly that command line (interpreter start + import torch + library load + masters read + one frame);
  (b) one process for all frames: `ap_calibrate.py - MBIAS MDARK - --batch LIST` (ApCalibrate.calibrate_files);
  (c) the phase breakdown of (b) from inside the process: FITS read + device decode, compute, encode + FITS write.
Beside it the reference's own 5.9 s per 4096^2 frame (its CPU path, stated as the baseline, not a target).

    python tools/bench_frame_files.py [--size 4096] [--frames 8] [--dir /tmp/apframes]
"""
import argparse
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--size', type=int, default=4096)
    ap.add_argument('--frames', type=int, default=8)
    ap.add_argument('--dir', default='/tmp/apframes')
    a = ap.parse_args()
    import numpy as np
    import torch
    from astrophotography_amd import fitsio, ops, synth
    import astrophotography_amd as apkg
    H = W = a.size
    d = a.dir
    os.makedirs(d, exist_ok=True)
    masters = synth.make_masters(H, W, config_id=2, device='cuda')
    nflat, _ = ops.flat_normalize(masters['flat'])
    frames = synth.make_frames(a.frames, masters, nflat, config_id=2, dtype=torch.uint16, first_frame=0)
    def hdr(**kw):
        h = fitsio.Header()
        for k, v in kw.items():
            h[k] = v
        return h

    fitsio.write(os.path.join(d, 'mbias.fits'), masters['bias'].cpu().numpy(), hdr(IMAGETYP='MASTER BIAS'), overwrite=True)
    fitsio.write(os.path.join(d, 'mdark.fits'), masters['dark'].cpu().numpy(), hdr(IMAGETYP='MASTER DARK', EXPTIME=300.0), overwrite=True)
    fitsio.write(os.path.join(d, 'mflat.fits'), masters['flat'].cpu().numpy(), hdr(IMAGETYP='MASTER FLAT'), overwrite=True)
    st = ops.sigclip_global(masters['dark'], sigma=4.0, maxiters=5)
    badmask, _ = ops.threshold_mask(masters['dark'], thresholds=st[3:5].contiguous())
    fitsio.write(os.path.join(d, 'mbadpix.fits'), badmask.cpu().numpy(), hdr(IMAGETYP='BADPIX'), overwrite=True)
    raws, cals = [], []
    host = frames.view(torch.int16).cpu().numpy().view(np.uint16)
    for i in range(a.frames):
        raws.append(os.path.join(d, 'raw%03d.fits' % i))
        cals.append(os.path.join(d, 'cal%03d.fits' % i))
        fitsio.write(raws[-1], host[i], hdr(IMAGETYP='Light Frame', EXPTIME=120.0, EGAIN=1.3), overwrite=True)
    del frames, masters
    torch.cuda.empty_cache()
    script = os.path.join(ROOT, 'astrophotography_amd', 'scripts', 'ap_calibrate.py')
    env = dict(os.environ, PYTHONPATH=ROOT + os.pathsep + os.environ.get('PYTHONPATH', ''))
    common = ['--master_flat=' + os.path.join(d, 'mflat.fits'), '--master_badpix=' + os.path.join(d, 'mbadpix.fits'), '--loglevel=ERROR']
    print('frames: %d x %d x %d uint16 FITS (%.1f MB each) in %s' % (a.frames, H, W, H * W * 2 / 1e6, d))
    for cosmic in (False, True):
        extra = common + (['--fixcosmic'] if cosmic else [])
        tag = 'with --fixcosmic' if cosmic else 'bias / dark / flat / bad pixels'
        # (a) one process per frame
        n_a = min(a.frames, 4)
        t0 = time.perf_counter()
        for i in range(n_a):
            subprocess.run([sys.executable, script, raws[i], os.path.join(d, 'mbias.fits'), os.path.join(d, 'mdark.fits'), cals[i]] + extra,
                           check=True, env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        ta = (time.perf_counter() - t0) / n_a
        # (b) one process, --batch
        lst = os.path.join(d, 'batch.txt')
        with open(lst, 'w') as fh:
            for r, c in zip(raws, cals):
                fh.write('%s %s\n' % (r, c))
        t0 = time.perf_counter()
        subprocess.run([sys.executable, script, '-', os.path.join(d, 'mbias.fits'), os.path.join(d, 'mdark.fits'), '-', '--batch', lst] + extra,
                       check=True, env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        tb = time.perf_counter() - t0
        # (c) inside one process (this one: interpreter, torch and the library are up)
        t0 = time.perf_counter()
        cal = apkg.ApCalibrate(os.path.join(d, 'mbias.fits'), os.path.join(d, 'mdark.fits'), os.path.join(d, 'mflat.fits'),
                               os.path.join(d, 'mbadpix.fits'), 'ERROR', False)
        torch.cuda.synchronize()
        t_masters = time.perf_counter() - t0
        tm = {}
        cal.calibrate_files(raws, cals, 2, cosmic, timings=tm)
        tm2 = {}
        cal.calibrate_files(raws, cals, 2, cosmic, timings=tm2)                      # second pass: page cache warm, kernels loaded
        print('== %s' % tag)
        print('(a) one process per frame (calibrate_all.sh pattern): %.3f s per frame = %.2f frames/s' % (ta, 1.0 / ta))
        print('(b) one process, --batch (%d frames):                  %.3f s in all = %.3f s per frame = %.2f frames/s' % (a.frames, tb, tb / a.frames, a.frames / tb))
        print('(c) inside a live process: masters %.3f s once; per frame %.1f ms = read+decode %.1f + compute %.1f + encode+write %.1f  -> %.1f frames/s'
              % (t_masters, 1e3 * tm2['total'] / a.frames, 1e3 * tm2['read'] / a.frames, 1e3 * tm2['compute'] / a.frames, 1e3 * tm2['write'] / a.frames,
                 a.frames / tm2['total']))
        print('    (first pass of (c): per frame %.1f ms = %.1f + %.1f + %.1f)' % (1e3 * tm['total'] / a.frames, 1e3 * tm['read'] / a.frames,
                                                                                   1e3 * tm['compute'] / a.frames, 1e3 * tm['write'] / a.frames))
    print('reference (BASELINE.md section 2, its CPU path on the survey container): 5.9 s per 4096^2 frame = 0.17 frames/s, without --fixcosmic')


if __name__ == '__main__':
    main()
