#!/opt/conda/bin/python3.9 -B
"""Golden-vector generator for the calibrate-and-stack hot path.

RUN ONLY IN THE BUILD CONTAINER:

    /opt/conda/bin/python3.9 -B tests/golden/make_golden.py

It imports the *reference* (``/root/reference``, AstroPhotography 0.5.1) together with the
astropy 4.3.1 / numpy 1.26.4 found in ``/opt/conda`` and records inputs + the reference's outputs
as small ``.npz`` / ``.fits`` fixtures next to this script.  Nothing of the reference's source is
copied: the fixtures hold arrays, header keyword values and versions only.  The GPU box never runs
this script (it has neither astropy nor the reference); tests read the committed fixtures.

Bootstrap recipe: SURVEY.md Appendix B (numpy alias shims for astropy 4.3.1, stub modules for the
third-party packages only the out-of-scope classes need, bottleneck disabled because the
reference's environment does not install it - requirements.txt:5-27).

Fixture groups:
  G1 calibrate      ApCalibrate.calibrate          core/ApCalibrate.py:406-509
  G2 find-badpix    ApFindBadPixels                core/ApFindBadPixels.py:30-68,171-217,414-438
  G3 fix-badpix     ApFixBadPixels.fix_bad_pixels  core/ApFixBadPixels.py:292-445
  G4 imarith        ApImArith.process_files        core/ApImArith.py:255-346
  G5 stack          astropy.stats.sigma_clipped_stats(axis=0) / sigma_clip(return_bounds=True)
  G6 mad_std/median along N (building blocks of the ccdproc.combine settings used at
                    scripts/ap_combine_darks.py:394-420; ccdproc itself is absent -> unpinned)
  G7 nanmean        np.nanmean of float32 flats (pairwise float32 summation) for _generate_flat
  G8 read noise     ApImageDifference / ApCalcReadNoise      scripts/ap_calc_read_noise.py:86-688
  G10 TAN WCS       astropy.wcs.WCS pixel <-> sky for RA---TAN / DEC--TAN headers (CD matrix, CDELT + CROTA2,
                    CDELT + PC): pins astrophotography_amd/wcs.py, the registration input of the resample step
                    (scripts/resample_all.sh:123-131 PROJECTION_TYPE TAN)
  G9 Bayer stamps   the known-answer tables of the reference's own RawConv.split test
                    (test/AstroPhotography/test_core.py:47-259: 14x14 R/G1/B/G2 stamps, with and without
                    black-level subtraction) - numbers only, read by calling the test class's accessor
"""
import sys, types, importlib, warnings, os, tempfile, shutil, json
warnings.filterwarnings('ignore')
import numpy as np

for nm, fn in [('asscalar', lambda a: a.item()), ('alen', len), ('msort', lambda a: np.sort(a, axis=0)),
               ('product', np.prod), ('cumproduct', np.cumprod), ('sometrue', np.any), ('alltrue', np.all),
               ('float', float), ('int', int), ('bool', bool), ('object', object),
               ('complex', complex), ('str', str)]:
    if not hasattr(np, nm):
        setattr(np, nm, fn)


class _Stub(types.ModuleType):
    def __getattr__(self, k):
        if k.startswith('__'):
            raise AttributeError(k)
        return _Stub(self.__name__ + '.' + k)


for m in ['rawpy', 'exifread', 'ccdproc', 'photutils', 'photutils.segmentation', 'photutils.background',
          'regions', 'astroquery', 'astroquery.astrometry_net', 'astroquery.exceptions',
          'astroplan', 'astroscrappy']:
    try:
        importlib.import_module(m)
    except Exception:
        sys.modules[m] = _Stub(m)

import astropy
import astropy.stats.sigma_clipping as sc
sc.HAS_BOTTLENECK = False
from astropy.io import fits
from astropy.stats import sigma_clipped_stats, sigma_clip, mad_std

sys.path.insert(0, '/root/reference')
import AstroPhotography as ap

HERE = os.path.dirname(os.path.abspath(__file__))
VERSIONS = dict(reference=ap.__version__, astropy=astropy.__version__, numpy=np.__version__,
                python=sys.version.split()[0], bottleneck='disabled')
LOG = 'CRITICAL'


def save(name, **arrs):
    arrs['_versions'] = np.array(json.dumps(VERSIONS))
    np.savez_compressed(os.path.join(HERE, name), **arrs)
    print('wrote', name, len(arrs) - 1, 'arrays')


def wfits(path, data, **kw):
    hdu = fits.PrimaryHDU(data)
    for k, v in kw.items():
        hdu.header[k.replace('_', '-')] = v
    hdu.writeto(path, overwrite=True)


def hdr_to_items(hdr):
    """Header -> list of [key, repr(value), comment] (HISTORY values recorded verbatim)."""
    out = []
    for c in hdr.cards:
        out.append([c.keyword, repr(c.value), c.comment])
    return json.dumps(out)


# ------------------------------------------------------------------------------------------------
def synth_masters(rng, H, W):
    bias = rng.normal(1000, 5, (H, W)).astype(np.float32)
    dark = rng.normal(20, 3, (H, W)).astype(np.float32)
    hot = rng.random((H, W)) < 0.002
    dark[hot] = rng.uniform(2000, 6000, hot.sum()).astype(np.float32)
    yy, xx = np.mgrid[0:H, 0:W]
    r2 = ((yy - H / 2) ** 2 + (xx - W / 2) ** 2) / (H * H / 4 + W * W / 4)
    flat = (rng.normal(30000, 300, (H, W)) * (1 - 0.3 * r2)).astype(np.float32)
    return bias, dark, flat


def g1_calibrate(tmp):
    rng = np.random.default_rng(101)
    cases = {}
    idx = 0
    for (H, W) in [(64, 64), (96, 80)]:
        bias, dark, flat = synth_masters(rng, H, W)
        flat_special = flat.copy()
        flat_special[3, 5] = 0.0
        flat_special[7, 9] = np.nan
        sky = (500 + 50 * np.arange(W)[None, :] / W) * np.ones((H, 1))
        rawf = bias + 0.4 * dark + (flat / flat.mean()) * sky + rng.normal(0, 12, (H, W))
        raw_u16 = np.clip(np.rint(rawf), 0, 65535).astype(np.uint16)
        raw_f32 = rawf.astype(np.float32)
        mask = np.zeros((H, W), np.uint8)
        mask[rng.random((H, W)) < 0.004] = 1
        mask[0, 0] = 1
        mask[H - 1, W - 1] = 3
        mask[10:15, 20:25] = 2      # 5x5 all-bad cluster
        for raw, rawname in [(raw_u16, 'u16'), (raw_f32, 'f32')]:
            for still_biased in [False, True]:
                for ped in [None, -100.0]:
                    for flatmode in ['flat', 'flat_special', 'noflat']:
                        for expkw in ['EXPTIME', 'EXPOSURE']:
                            for use_mask in [False, True]:
                                # Prune the cross product: keep a representative subset.
                                key = (rawname, still_biased, ped, flatmode, expkw, use_mask)
                                keep = (
                                    (flatmode == 'flat' and expkw == 'EXPTIME' and not use_mask) or
                                    (flatmode == 'flat_special' and rawname == 'u16' and not still_biased
                                     and ped is None and expkw == 'EXPOSURE') or
                                    (flatmode == 'noflat' and rawname == 'f32' and still_biased
                                     and ped is None and expkw == 'EXPTIME' and not use_mask) or
                                    (flatmode == 'flat' and rawname == 'u16' and not still_biased
                                     and ped is None and expkw == 'EXPTIME' and use_mask))
                                if not keep:
                                    continue
                                d = os.path.join(tmp, f'c{idx}')
                                os.makedirs(d, exist_ok=True)
                                wfits(f'{d}/bias.fits', bias)
                                wfits(f'{d}/dark.fits', dark, EXPTIME=300.0)
                                fl = {'flat': flat, 'flat_special': flat_special, 'noflat': None}[flatmode]
                                if fl is not None:
                                    wfits(f'{d}/flat.fits', fl)
                                kw = {expkw: 120.0}
                                if expkw == 'EXPOSURE':
                                    kw['EXPTIME'] = 999.0     # EXPOSURE must win (ApCalibrate.py:137)
                                if ped is not None:
                                    kw['PEDESTAL'] = ped
                                wfits(f'{d}/raw.fits', raw, **kw)
                                if use_mask:
                                    wfits(f'{d}/bpix.fits', mask)
                                cal = ap.ApCalibrate(f'{d}/bias.fits', f'{d}/dark.fits',
                                                     f'{d}/flat.fits' if fl is not None else None,
                                                     f'{d}/bpix.fits' if use_mask else None,
                                                     LOG, dark_still_biased=still_biased)
                                cal.calibrate(f'{d}/raw.fits', f'{d}/cal.fits', 2, None, False)
                                with fits.open(f'{d}/cal.fits') as hl:
                                    out = hl[0].data.copy()
                                    ohdr = hdr_to_items(hl[0].header)
                                pre = f'c{idx}_'
                                # inputs are stored once per shape: '<name>_<H>x<W>'
                                shp = f'_{H}x{W}'
                                cases['raw_' + rawname + shp] = raw
                                cases['bias' + shp] = bias
                                cases['dark' + shp] = dark
                                cases['mask' + shp] = mask
                                if fl is not None:
                                    cases[flatmode + shp] = fl
                                    cases['n' + flatmode + shp] = cal._norm_flat
                                cases[pre + 'out'] = out
                                cases[pre + 'meta'] = np.array(json.dumps(dict(
                                    raw=rawname, dark_still_biased=still_biased, pedestal=ped,
                                    flatmode=flatmode, expkw=expkw, use_mask=use_mask, shape=[H, W],
                                    img_exp=120.0, dark_exp=300.0, deltapix=2,
                                    out_dtype=str(out.dtype))))
                                cases[pre + 'hdr'] = np.array(ohdr)
                                if idx == 0:
                                    # keep one complete FITS in/out set for the FITS-I/O parity test
                                    for fn in ['raw', 'bias', 'dark', 'flat', 'cal']:
                                        shutil.copy(f'{d}/{fn}.fits', os.path.join(HERE, f'g1_c0_{fn}.fits'))
                                idx += 1
    cases['ncases'] = np.array(idx)
    save('g1_calibrate.npz', **cases)


def g2_findbadpix(tmp):
    rng = np.random.default_rng(202)
    out = {}
    for ci, (H, W, dt) in enumerate([(256, 256, np.float32), (300, 500, np.float32), (128, 128, np.uint16)]):
        dark = rng.normal(20, 3, (H, W))
        if dt == np.uint16:
            dark = dark + 100
        hot = rng.random((H, W)) < 0.0004
        dark[hot] = rng.uniform(2000, 6000, hot.sum())
        cold = rng.random((H, W)) < 0.0001
        dark[cold] = -300 if dt == np.float32 else 0
        dark = dark.astype(dt) if dt == np.float32 else np.clip(np.rint(dark), 0, 65535).astype(dt)
        p = f'{tmp}/dark{ci}.fits'
        wfits(p, dark, TELESCOP='synth', INSTRUME='cam', XBINNING=1)
        fb = ap.ApFindBadPixels(p, 4.0, LOG)
        mean, med, std = sigma_clipped_stats(fb._imdata, sigma=4.0)
        _, lo, hi = sc.SigmaClip(sigma=4.0)(fb._imdata, masked=False, return_bounds=True) \
            if False else (None, None, None)
        out[f'd{ci}_dark'] = dark
        out[f'd{ci}_mask_auto'] = fb.get_mask().copy()
        out[f'd{ci}_stats'] = np.array([float(mean), float(med), float(std)], np.float64)
        out[f'd{ci}_stats_dtype'] = np.array(str(np.asarray(mean).dtype))
        out[f'd{ci}_thresh'] = np.array([float(med - 4.0 * std), float(med + 4.0 * std)], np.float64)
        out[f'd{ci}_nbad_auto'] = np.array(int(fb._nbad_auto))
        if H >= 300:
            fb.add_user_badpix('/root/reference/etc/user_badpixels.yml')
            out[f'd{ci}_mask_user'] = fb.get_mask().copy()
            out[f'd{ci}_nbad_user'] = np.array(int(fb._nbad_user))
            mp = f'{tmp}/mask{ci}.fits'
            fb.write_mask(mp)
            with fits.open(mp) as hl:
                out[f'd{ci}_maskfile_hdr'] = np.array(hdr_to_items(hl[0].header))
                out[f'd{ci}_maskfile_dtype'] = np.array(str(hl[0].data.dtype))
    out['ncases'] = np.array(3)
    save('g2_findbadpix.npz', **out)


def g3_fixbadpix(tmp):
    rng = np.random.default_rng(303)
    out = {}
    H, W = 48, 40
    data = rng.normal(500, 20, (H, W)).astype(np.float32)
    mask = np.zeros((H, W), np.uint8)
    mask[rng.random((H, W)) < 0.02] = 1
    for (r, c) in [(0, 0), (0, W - 1), (H - 1, 0), (H - 1, W - 1), (0, 7), (13, 0), (H - 1, 20), (5, W - 1)]:
        mask[r, c] = 2
    mask[20:25, 10:15] = 1          # 5x5 all-bad cluster (centre unfixable for delta 1 and 2)
    mask[30:33, 30:33] = 3          # 3x3 cluster: centre unfixable for delta=1, fixable for delta=2
    mask[40, 5:9] = 1               # run of 4 in a row
    fx = ap.ApFixBadPixels(LOG)
    out['data'] = data
    out['mask'] = mask
    for dp in [1, 2, 3]:
        nd, st = fx.fix_bad_pixels(data, mask, dp)
        out[f'out_dp{dp}'] = nd
        out[f'stats_dp{dp}'] = np.array(json.dumps({k: [v[0].item() if hasattr(v[0], 'item') else v[0], v[1]]
                                                      for k, v in st.items()}))
    # integer input (reference warns but proceeds; medians truncate on assignment)
    datai = np.clip(np.rint(data), 0, 65535).astype(np.uint16)
    nd, st = fx.fix_bad_pixels(datai, mask, 1)
    out['data_u16'] = datai
    out['out_u16_dp1'] = nd
    # empty mask
    nd, st = fx.fix_bad_pixels(data, np.zeros_like(mask), 1)
    out['out_emptymask'] = nd
    out['stats_emptymask'] = np.array(json.dumps({k: [v[0].item() if hasattr(v[0], 'item') else v[0], v[1]]
                                                   for k, v in st.items()}))
    save('g3_fixbadpix.npz', **out)


def g4_imarith(tmp):
    rng = np.random.default_rng(404)
    out = {}
    H, W = 32, 48
    a = rng.normal(500, 50, (H, W)).astype(np.float32)
    b = rng.normal(10, 5, (H, W)).astype(np.float32)
    b[2, 3] = 0.0
    au = rng.integers(0, 65535, (H, W)).astype(np.uint16)
    bu = rng.integers(0, 65535, (H, W)).astype(np.uint16)
    wfits(f'{tmp}/a.fits', a, BUNIT='adu')
    wfits(f'{tmp}/b.fits', b)
    wfits(f'{tmp}/au.fits', au)
    wfits(f'{tmp}/bu.fits', bu)
    out.update(a=a, b=b, au=au, bu=bu)
    ia = ap.ApImArith(LOG)
    for op in ['ADD', 'SUB', 'MUL', 'DIV']:
        ia.process_files(f'{tmp}/a.fits', op, f'{tmp}/b.fits', f'{tmp}/o.fits', None)
        with fits.open(f'{tmp}/o.fits') as hl:
            out[f'f32_arr_{op}'] = hl[0].data.copy()
            if op == 'SUB':
                out['f32_arr_SUB_hdr'] = np.array(hdr_to_items(hl[0].header))
        ia.process_files(f'{tmp}/a.fits', ' ' + op.lower() + ' ', '3.25', f'{tmp}/o.fits', 'electrons')
        with fits.open(f'{tmp}/o.fits') as hl:
            out[f'f32_scl_{op}'] = hl[0].data.copy()
            if op == 'MUL':
                out['f32_scl_MUL_hdr'] = np.array(hdr_to_items(hl[0].header))
    for op in ['ADD', 'SUB', 'MUL']:
        try:
            ia.process_files(f'{tmp}/au.fits', op, f'{tmp}/bu.fits', f'{tmp}/o.fits', None)
            with fits.open(f'{tmp}/o.fits', uint=True) as hl:
                out[f'u16_arr_{op}'] = hl[0].data.copy()
            out[f'u16_arr_{op}_exc'] = np.array('')
        except Exception as e:
            out[f'u16_arr_{op}_exc'] = np.array(type(e).__name__)
    for tag, args in [('u16_arr_DIV', ('DIV', f'{tmp}/bu.fits')), ('u16_scl_ADD', ('ADD', '2.0')),
                      ('f32_badop', ('POW', '2.0')), ('f32_badfile', ('ADD', f'{tmp}/nonexistent.fits'))]:
        try:
            src = f'{tmp}/au.fits' if tag.startswith('u16') else f'{tmp}/a.fits'
            ia.process_files(src, args[0], args[1], f'{tmp}/o.fits', None)
            out[tag + '_exc'] = np.array('')
        except Exception as e:
            out[tag + '_exc'] = np.array(type(e).__name__)
    save('g4_imarith.npz', **out)


def g5_stack(tmp):
    rng = np.random.default_rng(505)
    out = {}
    ci = 0
    H, W = 12, 16
    for N in [3, 8, 16, 64]:
        cube = rng.normal(500, 20, (N, H, W)).astype(np.float32)
        # cosmic-ray like outliers
        hits = rng.random((N, H, W)) < 0.03
        cube[hits] += rng.uniform(200, 5000, hits.sum()).astype(np.float32)
        lows = rng.random((N, H, W)) < 0.01
        cube[lows] -= rng.uniform(200, 400, lows.sum()).astype(np.float32)
        cube[:, 0, 0] = 123.0                      # all-equal column (std = 0)
        cube[:, 0, 1] = np.nan                     # all-NaN column
        cube[0, 0, 2] = np.nan                     # one NaN
        cube[min(1, N - 1), 0, 3] = np.inf         # one +inf
        cube[min(2, N - 1), 0, 4] = -np.inf
        cube[:, 0, 5] = np.arange(N, dtype=np.float32)      # ramp
        cube[:, 0, 6] = 7.0
        cube[0, 0, 6] = 9.0                        # all equal but one
        out[f'cube_N{N}'] = cube
        for sigma in [3.0, 5.0]:
            for maxiters in [1, 5, None]:
                for cen in ['median', 'mean']:
                    for sd in ['std', 'mad_std']:
                        if sd == 'mad_std' and not (sigma == 5.0 or maxiters == 5):
                            continue
                        mean, med, std = sigma_clipped_stats(cube, sigma=sigma, maxiters=maxiters,
                                                             cenfunc=cen, stdfunc=sd, axis=0)
                        filt, lo, hi = sigma_clip(cube, sigma=sigma, maxiters=maxiters, cenfunc=cen,
                                                  stdfunc=sd, axis=0, masked=True, return_bounds=True)
                        p = f's{ci}_'
                        out[p + 'cfg'] = np.array(json.dumps(dict(N=N, sigma=sigma, maxiters=maxiters,
                                                                  cenfunc=cen, stdfunc=sd)))
                        out[p + 'mean'] = np.asarray(mean, np.float64)
                        out[p + 'median'] = np.asarray(med, np.float64)
                        out[p + 'std'] = np.asarray(std, np.float64)
                        out[p + 'lo'] = np.asarray(lo, np.float64)
                        out[p + 'hi'] = np.asarray(hi, np.float64)
                        out[p + 'mask'] = np.packbits(np.asarray(filt.mask, bool))
                        ci += 1
    # asymmetric sigma_lower / sigma_upper (the ccdproc-style call uses separate low/high thresholds)
    cube = out['cube_N16']
    mean, med, std = sigma_clipped_stats(cube, sigma_lower=2.0, sigma_upper=4.0, maxiters=5, axis=0)
    out['asym_mean'] = np.asarray(mean, np.float64)
    out['asym_cfg'] = np.array(json.dumps(dict(N=16, sigma_lower=2.0, sigma_upper=4.0, maxiters=5,
                                                cenfunc='median', stdfunc='std')))
    # u16 cube (astropy converts non-float input to float64 before the C loop)
    cu = np.clip(np.rint(out['cube_N8'][:, 1:, :]), 0, 65535).astype(np.uint16)
    mean, med, std = sigma_clipped_stats(cu, sigma=3.0, maxiters=5, axis=0)
    out['u16_cube'] = cu
    out['u16_mean'] = np.asarray(mean, np.float64)
    out['u16_median'] = np.asarray(med, np.float64)
    out['u16_std'] = np.asarray(std, np.float64)
    out['ncfg'] = np.array(ci)
    save('g5_stack.npz', **out)


def g6_madstd(tmp):
    rng = np.random.default_rng(606)
    out = {}
    for N in [5, 8, 16]:
        cube = rng.normal(100, 10, (N, 10, 12)).astype(np.float32)
        hits = rng.random(cube.shape) < 0.05
        cube[hits] += 500
        out[f'cube_N{N}'] = cube
        out[f'median_N{N}'] = np.median(cube.astype(np.float64), axis=0)
        out[f'mamedian_N{N}'] = np.ma.median(np.ma.masked_invalid(cube.astype(np.float64)), axis=0).filled(np.nan)
        out[f'madstd_N{N}'] = mad_std(cube.astype(np.float64), axis=0)
    save('g6_madstd.npz', **out)


def g7_nanmean(tmp):
    rng = np.random.default_rng(707)
    out = {}
    for ci, shape in enumerate([(5,), (7, 1), (3, 37), (64, 64), (96, 80), (129, 1), (300, 500), (1, 1000), (513, 257)]):
        a = (rng.normal(30000, 300, shape)).astype(np.float32)
        out[f'a{ci}'] = a
        out[f'nanmean{ci}'] = np.array(np.nanmean(a))
        out[f'nanmean{ci}_dtype'] = np.array(str(np.asarray(np.nanmean(a)).dtype))
        out[f'sum{ci}'] = np.array(np.sum(a))
        b = a.copy()
        b.flat[::17] = np.nan
        out[f'nanmean_withnan{ci}'] = np.array(np.nanmean(b))
        # the numerics the global clip relies on (ApFindBadPixels.py:191 via numpy nan-functions)
        out[f'nanstd{ci}'] = np.array(np.nanstd(a))
        out[f'nanmedian{ci}'] = np.array(np.nanmedian(a))
    out['ncases'] = np.array(9)
    save('g7_nanmean.npz', **out)


def g8_readnoise(tmp):
    import importlib.util
    import matplotlib
    matplotlib.use('Agg')
    spec = importlib.util.spec_from_file_location('ref_ap_calc_read_noise',
                                                  '/root/reference/AstroPhotography/scripts/ap_calc_read_noise.py')
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    rng = np.random.default_rng(808)
    out = {}
    H, W = 120, 200
    for tag, dt in (('u16', np.uint16), ('f32', np.float32)):
        b1 = rng.normal(1000, 8, (H, W))
        b2 = rng.normal(1000, 8, (H, W))
        hot = rng.random((H, W)) < 0.002
        b1[hot] += 3000
        b2[rng.random((H, W)) < 0.002] += 2500
        if dt == np.uint16:
            b1 = np.clip(np.rint(b1), 0, 65535).astype(dt)
            b2 = np.clip(np.rint(b2), 0, 65535).astype(dt)
        else:
            b1, b2 = b1.astype(dt), b2.astype(dt)
        out[tag + '_b1'], out[tag + '_b2'] = b1, b2
        for clip in (True, False):
            d = mod.ApImageDifference(b1, b2, clip, LOG)
            ng, nt = d.numpix()
            out[f'{tag}_clip{int(clip)}_stats'] = np.array([d.stddev(), d.min(), d.max(), d.mean(), d.median(), ng, nt], np.float64)
            out[f'{tag}_clip{int(clip)}_good'] = np.packbits(d.good_pixel_mask())
        m1 = (rng.random((H, W)) < 0.01).astype(np.uint8) * 3
        d = mod.ApImageDifference(b1, b2, False, LOG, mask1=m1, mask2=None)
        out[tag + '_mask1'] = m1
        out[tag + '_masked_stats'] = np.array([d.stddev(), d.min(), d.max(), d.mean(), d.median(), d.numpix()[0]], np.float64)
        p1, p2 = f'{tmp}/rn_{tag}_1.fits', f'{tmp}/rn_{tag}_2.fits'
        wfits(p1, b1, EGAIN=1.37)
        wfits(p2, b2, EGAIN=1.37)
        rn = mod.ApCalcReadNoise(p1, p2, 'EGAIN', LOG).estimate_rn(True)
        rn2 = mod.ApCalcReadNoise(p1, p2, '2.0', LOG).estimate_rn(False)
        out[tag + '_readnoise'] = np.array([rn, rn2], np.float64)
    save('g8_readnoise.npz', **out)


def g9_bayer_stamps(tmp):
    """The reference's hand-checked split() stamps (the CR2 they were cut from is not in the checkout)."""
    sys.path.insert(0, '/root/reference/test/AstroPhotography')
    import importlib.util
    spec = importlib.util.spec_from_file_location('ref_test_core', '/root/reference/test/AstroPhotography/test_core.py')
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    t = mod.RawConvTest()
    out = dict(versions=json.dumps(VERSIONS), channels=json.dumps(['R', 'G1', 'B', 'G2']))
    for black in (False, True):
        for ch in ('R', 'G1', 'B', 'G2'):
            oct_ind, meanval, stdval, minval, maxval, sumval, img = t._get_split_data(ch, black)
            tag = '%s_%s' % (ch, 'black' if black else 'noblack')
            out[tag] = np.asarray(img, np.int64)
            out[tag + '_octind'] = np.asarray(oct_ind, np.int64)
            out[tag + '_stats'] = np.asarray([meanval, stdval, minval, maxval, sumval], np.int64)
    save('g9_bayer_stamps.npz', **out)


def g10_wcs(tmp):
    from astropy.wcs import WCS
    rng = np.random.default_rng(1010)
    cases = [
        dict(CTYPE1='RA---TAN', CTYPE2='DEC--TAN', CRPIX1=960.5, CRPIX2=540.5, CRVAL1=303.0272359, CRVAL2=38.3549333,
             CD1_1=-4.9e-4, CD1_2=1.2e-5, CD2_1=1.3e-5, CD2_2=4.95e-4),
        dict(CTYPE1='RA---TAN', CTYPE2='DEC--TAN', CRPIX1=2048.0, CRPIX2=2048.0, CRVAL1=10.68, CRVAL2=-41.27,
             CDELT1=-2.5e-4, CDELT2=2.5e-4, CROTA2=12.5),
        dict(CTYPE1='RA---TAN', CTYPE2='DEC--TAN', CRPIX1=100.25, CRPIX2=-20.5, CRVAL1=359.9, CRVAL2=85.0,
             CDELT1=-5e-4, CDELT2=5e-4, PC1_1=0.8, PC1_2=-0.6, PC2_1=0.6, PC2_2=0.8),
    ]
    out = dict(versions=json.dumps(VERSIONS), ncases=len(cases))
    for i, c in enumerate(cases):
        h = fits.Header()
        for k, v in c.items():
            h[k] = v
        w = WCS(h)
        pix = np.column_stack([rng.uniform(-50, 4200, 200), rng.uniform(-50, 4200, 200)])
        sky = w.all_pix2world(pix, 0)
        back = w.all_world2pix(sky, 0)
        out['hdr%d' % i] = json.dumps(c)
        out['pix%d' % i] = pix
        out['sky%d' % i] = sky
        out['back%d' % i] = back
    save('g10_wcs.npz', **out)


if __name__ == '__main__':
    tmp = tempfile.mkdtemp(prefix='apgold_')
    try:
        which = sys.argv[1:] or ['g1', 'g2', 'g3', 'g4', 'g5', 'g6', 'g7', 'g8', 'g9', 'g10']
        for g in which:
            {'g1': g1_calibrate, 'g2': g2_findbadpix, 'g3': g3_fixbadpix, 'g4': g4_imarith,
             'g5': g5_stack, 'g6': g6_madstd, 'g7': g7_nanmean, 'g8': g8_readnoise, 'g9': g9_bayer_stamps, 'g10': g10_wcs}[g](tmp)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
