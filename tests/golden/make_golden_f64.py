#!/opt/conda/bin/python3.9 -B
"""Golden vectors for the float64 side of ApCalibrate (group G11).  RUN ONLY IN THE BUILD CONTAINER:

    /opt/conda/bin/python3.9 -B tests/golden/make_golden_f64.py

ApCalibrate._read_fits converts only NON-float data to float32 (core/ApCalibrate.py:301-305), and ApMasterCal
writes float64 masters (scripts/ap_combine_darks.py:437), so the reference calibrates with float64 masters in
float64 (NumPy type promotion per operation: raw - bias, dark - bias, e * D, x - e*D, flat / nanmean(flat),
x / nflat) and writes a BITPIX -64 image.  This script runs the imported reference on every dtype mix the
promotion rules distinguish and records inputs, outputs (with their dtype), the normalised flat and the
bad-pixel-fix statistics; plus np.nanmean / np.sum of float64 arrays (the summation tree of _generate_flat).
The bootstrap (numpy shims, stub modules, bottleneck off) is make_golden.py's, imported as a module.
"""
import json
import os
import shutil
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg          # noqa: E402  (bootstraps astropy + the reference; its __main__ block does not run)

ap, fits = mg.ap, mg.fits


def g11_calibrate_f64(tmp):
    rng = np.random.default_rng(1111)
    H, W = 96, 100                                   # 9600 pixels: more than one 8192-element summation piece
    bias32, dark32, flat32 = mg.synth_masters(rng, H, W)
    bias64 = bias32.astype(np.float64) + rng.normal(0, 0.01, (H, W))      # genuinely float64 values (a ccdproc mean)
    dark64 = dark32.astype(np.float64) + rng.normal(0, 0.01, (H, W))
    flat64 = flat32.astype(np.float64) + rng.normal(0, 0.01, (H, W))
    flat64s = flat64.copy()
    flat64s[3, 5] = 0.0
    flat64s[7, 9] = np.nan
    sky = (500 + 50 * np.arange(W)[None, :] / W) * np.ones((H, 1))
    rawf = bias64 + 0.4 * dark64 + (flat64 / flat64.mean()) * sky + rng.normal(0, 12, (H, W))
    raws = dict(u16=np.clip(np.rint(rawf), 0, 65535).astype(np.uint16), f32=rawf.astype(np.float32), f64=rawf)
    mask = np.zeros((H, W), np.uint8)
    mask[rng.random((H, W)) < 0.004] = 1
    mask[0, 0] = 1
    mask[H - 1, W - 1] = 3
    mask[10:15, 20:25] = 2
    masters = dict(bias32=bias32, bias64=bias64, dark32=dark32, dark64=dark64, flat32=flat32, flat64=flat64, flat64s=flat64s)
    #        raw    bias      dark      flat      still_biased pedestal mask
    plan = [('u16', 'bias64', 'dark64', 'flat64', False, None, True),      # masters made by ApMasterCal, raw camera frame
            ('u16', 'bias64', 'dark64', 'flat64s', True, None, False),     # biased dark, flat with 0 and NaN
            ('f32', 'bias64', 'dark32', 'flat32', False, None, False),     # only the bias is float64
            ('f32', 'bias32', 'dark64', 'flat32', False, None, False),     # only the dark is float64 (e stays a float64 scalar)
            ('f32', 'bias32', 'dark32', 'flat64', False, None, True),      # only the flat is float64: float32 chain, float64 division
            ('f64', 'bias32', 'dark32', 'flat32', False, -100.0, False),   # only the raw frame is float64, with PEDESTAL
            ('f64', 'bias64', 'dark64', None, True, None, False),          # all float64, no flat
            ('u16', 'bias32', 'dark64', None, True, None, False),          # D = dark64 - bias32
            ('f32', 'bias64', 'dark64', 'flat64', False, -100.0, False)]   # float32 raw + PEDESTAL (added in float32), f64 masters
    out = {}
    for k, v in list(raws.items()):
        out['raw_' + k] = v
    out.update(masters)
    out['mask'] = mask
    for idx, (rk, bk, dk, fk, sb, ped, use_mask) in enumerate(plan):
        d = os.path.join(tmp, f'f{idx}')
        os.makedirs(d, exist_ok=True)
        mg.wfits(f'{d}/bias.fits', masters[bk])
        mg.wfits(f'{d}/dark.fits', masters[dk], EXPTIME=300.0)
        if fk is not None:
            mg.wfits(f'{d}/flat.fits', masters[fk])
        kw = {'EXPTIME': 120.0}
        if ped is not None:
            kw['PEDESTAL'] = ped
        mg.wfits(f'{d}/raw.fits', raws[rk], **kw)
        if use_mask:
            mg.wfits(f'{d}/bpix.fits', mask)
        cal = ap.ApCalibrate(f'{d}/bias.fits', f'{d}/dark.fits', f'{d}/flat.fits' if fk else None,
                             f'{d}/bpix.fits' if use_mask else None, mg.LOG, dark_still_biased=sb)
        cal.calibrate(f'{d}/raw.fits', f'{d}/cal.fits', 2, None, False)
        with fits.open(f'{d}/cal.fits') as hl:
            res = hl[0].data.copy()
            ohdr = mg.hdr_to_items(hl[0].header)
            bitpix = int(hl[0].header['BITPIX'])
        res = res.astype(res.dtype.newbyteorder('='))
        pre = f'f{idx}_'
        out[pre + 'out'] = res
        if fk is not None:
            nf = np.asarray(cal._norm_flat)
            out[pre + 'nflat'] = nf.astype(nf.dtype.newbyteorder('='))
        out[pre + 'meta'] = np.array(json.dumps(dict(raw=rk, bias=bk, dark=dk, flat=fk, dark_still_biased=sb, pedestal=ped,
                                                     use_mask=use_mask, img_exp=120.0, dark_exp=300.0, deltapix=2,
                                                     out_dtype=str(res.dtype), bitpix=bitpix)))
        out[pre + 'hdr'] = np.array(ohdr)
        print(idx, rk, bk, dk, fk, '->', res.dtype, bitpix)
    out['ncases'] = np.array(len(plan))
    # np.nanmean / np.sum of float64 arrays: pins the float64 summation order _generate_flat depends on
    for j, n in enumerate([100, 8192, 8193, 9600, 20000]):
        a = rng.normal(30000, 300, n)
        b = a.copy()
        b[::97] = np.nan
        out[f's{j}_a'] = a
        out[f's{j}_sum'] = np.array(np.sum(a))
        out[f's{j}_nanmean'] = np.array(np.nanmean(b))
        out[f's{j}_mean2d'] = np.array(np.nanmean(a[:(n // 4) * 4].reshape(4, -1)))
    out['nsums'] = np.array(5)
    mg.save('g11_calibrate_f64.npz', **out)


if __name__ == '__main__':
    tmp = tempfile.mkdtemp(prefix='apgold64_')
    try:
        g11_calibrate_f64(tmp)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
