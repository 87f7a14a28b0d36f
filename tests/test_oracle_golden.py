"""Pins the CPU oracle (oracle/apref.c) against golden vectors captured from the imported reference
(tests/golden/make_golden.py: AstroPhotography 0.5.1 + astropy 4.3.1 + numpy 1.26.4).  CPU only."""
import json
import os

import numpy as np
import pytest

from oracle import apref


def load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name), allow_pickle=False)


def bits(a):
    a = np.ascontiguousarray(a)
    return a.view({4: np.uint32, 8: np.uint64, 2: np.uint16, 1: np.uint8}[a.dtype.itemsize])


def assert_biteq(a, b):
    a = np.asarray(a)
    b = np.asarray(b)
    assert a.dtype == b.dtype and a.shape == b.shape
    nan = np.isnan(a) & np.isnan(b) if a.dtype.kind == 'f' else np.zeros(a.shape, bool)
    ok = (bits(a) == bits(b)) | nan
    assert ok.all(), f'{(~ok).sum()} of {ok.size} differ; first {np.argwhere(~ok)[:3]}'


# ---- G7: numpy pairwise float32 summation / nanmean -------------------------------------------
def test_g7_pairwise_and_nanmean(golden_dir):
    g = load(golden_dir, 'g7_nanmean.npz')
    for ci in range(int(g['ncases'])):
        a = g[f'a{ci}']
        assert bits(apref.pairwise_sum_f32(a)) == bits(g[f'sum{ci}'])
        _, norm = apref.flat_normalize(a)
        assert str(g[f'nanmean{ci}_dtype']) == 'float32'
        assert bits(norm) == bits(g[f'nanmean{ci}']), ci
        b = a.copy()
        b.flat[::17] = np.nan
        _, normb = apref.flat_normalize(b)
        assert bits(normb) == bits(g[f'nanmean_withnan{ci}']), ci
        st = apref.sigclip_global(a, sigma=1e30, maxiters=1)       # nothing clipped: plain nanstd/median
        assert bits(st['std']) == bits(g[f'nanstd{ci}']), ci
        assert bits(st['median']) == bits(g[f'nanmedian{ci}']), ci
        assert bits(st['mean']) == bits(g[f'nanmean{ci}']), ci


# ---- G1: ApCalibrate ----------------------------------------------------------------------------
def test_g1_calibrate(golden_dir):
    g = load(golden_dir, 'g1_calibrate.npz')
    n = int(g['ncases'])
    assert n >= 20
    for ci in range(n):
        meta = json.loads(str(g[f'c{ci}_meta']))
        H, W = meta['shape']
        shp = f'_{H}x{W}'
        raw = g['raw_' + meta['raw'] + shp]
        bias, dark = g['bias' + shp], g['dark' + shp]
        nflat = None
        if meta['flatmode'] != 'noflat':
            nflat, norm = apref.flat_normalize(g[meta['flatmode'] + shp])
            assert_biteq(nflat, g['n' + meta['flatmode'] + shp])
        e = meta['img_exp'] / meta['dark_exp']
        out = apref.calibrate(raw, bias, dark, nflat, e, pedestal=meta['pedestal'],
                              dark_still_biased=meta['dark_still_biased'])
        if meta['use_mask']:
            out, st = apref.fix_badpix(out, g['mask' + shp], meta['deltapix'])
            hdr = {k: v for k, v, _ in json.loads(str(g[f'c{ci}_hdr']))}
            assert int(hdr['BPIXNBAD']) == st['nbad']
            assert int(hdr['BPIXNFIX']) == st['nfix']
            assert int(hdr['BPIXNREM']) == st['nrem']
        ref = g[f'c{ci}_out']
        assert meta['out_dtype'] in ('>f4', 'float32')
        assert_biteq(out, ref.astype(np.float32))


# ---- G2: ApFindBadPixels ------------------------------------------------------------------------
def test_g2_findbadpix(golden_dir):
    g = load(golden_dir, 'g2_findbadpix.npz')
    for ci in range(int(g['ncases'])):
        dark = g[f'd{ci}_dark']
        st = apref.sigclip_global(dark, sigma=4.0, maxiters=5)
        ref = g[f'd{ci}_stats']
        if dark.dtype == np.float32:
            assert str(g[f'd{ci}_stats_dtype']) == 'float32'
            for k, r in zip(('mean', 'median', 'std'), ref):
                assert bits(st[k]) == bits(np.float32(r)), (ci, k, st[k], r)
        else:
            for k, r in zip(('mean', 'median', 'std'), ref):
                assert st[k] == r, (ci, k, st[k], r)
        lo, hi = apref.badpix_thresholds(st['median'], st['std'], 4.0)
        assert [lo, hi] == list(g[f'd{ci}_thresh'])
        mask, nbad = apref.threshold_mask(dark, lo, hi)
        assert np.array_equal(mask, g[f'd{ci}_mask_auto'])
        assert nbad == int(g[f'd{ci}_nbad_auto'])
        assert nbad > 0
        if f'd{ci}_mask_user' in g:
            # etc/user_badpixels.yml: columns 12,13,17; rectangles [1,1,1,1],[5,6,7,12],[200,300,400,420]
            H, W = mask.shape
            rects = [[0, H, c - 1, c] for c in (12, 13, 17)]
            rects += [[0, 1, 0, 1], [4, 6, 6, 12], [199, 300, 399, 420]]
            m2 = apref.mask_add_rects(mask, rects, 2)
            assert np.array_equal(m2, g[f'd{ci}_mask_user'])
            assert m2.max() >= 4         # overlapping user regions sum
            assert sum((r1 - r0) * (c1 - c0) for r0, r1, c0, c1 in rects) == int(g[f'd{ci}_nbad_user'])


# ---- G3: ApFixBadPixels -------------------------------------------------------------------------
def test_g3_fixbadpix(golden_dir):
    g = load(golden_dir, 'g3_fixbadpix.npz')
    data, mask = g['data'], g['mask']
    for dp in (1, 2, 3):
        out, st = apref.fix_badpix(data, mask, dp)
        assert_biteq(out, g[f'out_dp{dp}'])
        ref = json.loads(str(g[f'stats_dp{dp}']))
        assert st['nbad'] == ref['BPIXNBAD'][0]
        assert st['nfix'] == ref['BPIXNFIX'][0]
        assert st['nrem'] == ref['BPIXNREM'][0]
        assert ref['BPIX_MIN'][0] == 4 and ref['BPIXDPIX'][0] == dp
    assert json.loads(str(g['stats_dp1']))['BPIXNREM'][0] > 0       # unfixable cluster centre exercised
    out, st = apref.fix_badpix(data, np.zeros_like(mask), 1)
    assert_biteq(out, g['out_emptymask'])
    assert st == dict(nbad=0, nfix=0, nrem=0)


# ---- G4: ApImArith ------------------------------------------------------------------------------
def test_g4_imarith(golden_dir):
    g = load(golden_dir, 'g4_imarith.npz')
    a, b, au, bu = g['a'], g['b'], g['au'], g['bu']
    for op in ('ADD', 'SUB', 'MUL', 'DIV'):
        with np.errstate(all='ignore'):
            assert_biteq(apref.imarith(a, op, b), g[f'f32_arr_{op}'].astype(np.float32))
            assert_biteq(apref.imarith(a, op, 3.25), g[f'f32_scl_{op}'].astype(np.float32))
    for op in ('ADD', 'SUB', 'MUL'):
        assert str(g[f'u16_arr_{op}_exc']) == ''
        assert np.array_equal(apref.imarith(au, op, bu), g[f'u16_arr_{op}'])
    # the reference's error behaviour, recorded for the host-side tests
    assert str(g['u16_arr_DIV_exc']) != '' and str(g['u16_scl_ADD_exc']) != ''
    assert str(g['f32_badop_exc']) == 'ValueError' and str(g['f32_badfile_exc']) == 'ValueError'


# ---- G5: sigma_clipped_stats(axis=0) ------------------------------------------------------------
def test_g5_stack(golden_dir):
    g = load(golden_dir, 'g5_stack.npz')
    ncfg = int(g['ncfg'])
    assert ncfg == 80
    worst = 0.0
    for ci in range(ncfg):
        cfg = json.loads(str(g[f's{ci}_cfg']))
        cube = g[f'cube_N{cfg["N"]}']
        r = apref.stack_sigclip(cube, sigma=cfg['sigma'], maxiters=cfg['maxiters'],
                                cenfunc=cfg['cenfunc'], stdfunc=cfg['stdfunc'])
        refmask = np.unpackbits(g[f's{ci}_mask'])[:cube.size].reshape(cube.shape).astype(bool)
        assert np.array_equal(~r['keep'], refmask), cfg
        for k in ('mean', 'median', 'std'):
            assert_biteq(r[k], g[f's{ci}_{k}'])
        for k in ('lo', 'hi'):
            ref = g[f's{ci}_{k}']
            both_nan = np.isnan(ref) & np.isnan(r[k])
            with np.errstate(invalid='ignore', divide='ignore'):
                rel = np.where(both_nan, 0.0, np.abs(r[k] - ref) / np.maximum(np.abs(ref), 1e-300))
            worst = max(worst, float(np.nanmax(rel)))
            assert np.array_equal(np.isnan(ref), np.isnan(r[k]))
    assert worst <= 1e-13, worst
    cfg = json.loads(str(g['asym_cfg']))
    r = apref.stack_sigclip(g['cube_N16'], sigma_lower=cfg['sigma_lower'], sigma_upper=cfg['sigma_upper'],
                            maxiters=cfg['maxiters'])
    assert_biteq(r['mean'], g['asym_mean'])
    r = apref.stack_sigclip(g['u16_cube'], sigma=3.0, maxiters=5)
    assert_biteq(r['mean'], g['u16_mean'])
    assert_biteq(r['median'], g['u16_median'])
    assert_biteq(r['std'], g['u16_std'])


# ---- G6: median / mad_std along N (building blocks of the ccdproc.combine configuration) --------
def test_g6_madstd_blocks(golden_dir):
    g = load(golden_dir, 'g6_madstd.npz')
    for N in (5, 8, 16):
        cube = g[f'cube_N{N}']
        assert_biteq(apref.stack_median(cube), g[f'median_N{N}'])
        assert_biteq(apref.stack_median(cube), g[f'mamedian_N{N}'])
        # mad_std = 1.4826 * median(|x - median|): exercised through the clip with sigma so large
        # nothing is rejected -> bounds = median -/+ sigma*mad_std
        r = apref.stack_sigclip(cube, sigma=1.0, maxiters=1, cenfunc='median', stdfunc='mad_std', want=('lo', 'hi'))
        med = g[f'median_N{N}']
        np.testing.assert_allclose((r['hi'] - r['lo']) / 2.0, g[f'madstd_N{N}'], rtol=1e-13)
        # the ccdproc-style one-pass combine is self-consistent with those blocks
        c = apref.combine_ccdproc(cube, 5.0, 5.0)
        sd = g[f'madstd_N{N}']
        keep = ~((cube - med < -5 * sd) | (cube - med > 5 * sd))
        exp = np.where(keep, cube.astype(np.float64), 0).sum(0) / keep.sum(0)
        np.testing.assert_allclose(c['mean'], exp, rtol=1e-14)
        assert np.array_equal(c['count'], keep.sum(0))


# ---- G12: the Combiner steps of ccdproc.combine, run with the third-party calls it makes ------------------
def test_g12_combine_ccdproc(golden_dir):
    """apref_combine_ccdproc_form against BOTH published forms of ccdproc's Combiner.sigma_clipping, each run with the third-party
    calls it makes (tests/golden/make_golden_combine.py; scripts/ap_combine_darks.py:394-420):
      'legacy'  (ccdproc <= 2.1) np.ma.median + astropy.stats.mad_std + differences against -low dev / high dev on the masked cube;
      'astropy' (ccdproc >= 2.2) astropy.stats.sigma_clip(maxiters=1, cenfunc=np.ma.median, stdfunc=mad_std) run for real.
    Means, survivor counts and stds bit for bit in both; values exactly on a +-5 dev bound kept (strict inequalities); where the
    forms disagree - columns holding a non-finite value (unclipped in the astropy form) and float64 values within an ulp of a
    bound with a non-zero base - the oracle follows each."""
    import json
    g = load(golden_dir, 'g12_combine.npz')
    ties = 0
    disagree = {'nonfinite': 0, 'f64bounds': 0}
    for m in json.loads(str(g['_meta'])):
        k = m['case']
        fr = g[f'c{k}_frames']
        cube = fr.astype(np.float32) if fr.dtype == np.uint16 else fr          # uint16 -> float32 -> float64 is exact
        r = apref.combine_ccdproc(cube, 5.0, 5.0, form='legacy')
        assert np.array_equal(r['count'], g[f'c{k}_count']), m
        assert_biteq(r['mean'], g[f'c{k}_mean'])
        assert_biteq(r['std'], g[f'c{k}_std'])
        rb = apref.combine_ccdproc(cube, 5.0, 5.0)                              # the default: 'astropy'
        assert np.array_equal(rb['count'], g[f'c{k}_b_count']), m
        assert_biteq(rb['mean'], g[f'c{k}_b_mean'])
        assert_biteq(rb['std'], g[f'c{k}_b_std'])
        differ = g[f'c{k}_count'] != g[f'c{k}_b_count']
        if m['kind'] == 'f64bounds':
            disagree['f64bounds'] += int(differ.sum())
        else:
            # on finite data the two forms keep the same values; they part only where a column holds NaN / inf
            nonfinite = ~np.isfinite(np.asarray(fr, dtype=np.float64)).all(axis=0)
            assert not (differ & ~nonfinite).any(), m
            disagree['nonfinite'] += int(differ.sum())
        if m['kind'] == 'f64ties' and m['N'] >= 8:
            col, base, dev = fr[:, 0, 0], g[f'c{k}_baseline'][0, 0], g[f'c{k}_dev'][0, 0]
            assert col[-2] - base == 5.0 * dev and col[-1] - base == -5.0 * dev      # exactly ON the bounds ...
            assert g[f'c{k}_count'][0, 0] == m['N'] and g[f'c{k}_b_count'][0, 0] == m['N']   # ... and kept by both forms
            ties += 1
    assert ties == 5
    assert disagree['nonfinite'] >= 6 and disagree['f64bounds'] >= 20, disagree


# ---- G8: ApImageDifference / ApCalcReadNoise -------------------------------------------------------
def test_g8_read_noise(golden_dir):
    import math
    g = load(golden_dir, 'g8_readnoise.npz')
    for tag in ('u16', 'f32'):
        b1, b2 = g[tag + '_b1'], g[tag + '_b2']
        for clip in (1, 0):
            r = apref.image_difference(b1, b2, bool(clip))
            ref = g[f'{tag}_clip{clip}_stats']
            assert [r['stddev'], r['min'], r['max'], r['mean'], r['median'], r['numgood'], r['numpix']] == list(ref), (tag, clip)
            good = np.unpackbits(g[f'{tag}_clip{clip}_good'])[:b1.size].reshape(b1.shape).astype(bool)
            assert np.array_equal(r['good'], good)
        r = apref.image_difference(b1, b2, False, mask1=g[tag + '_mask1'])
        assert [r['stddev'], r['min'], r['max'], r['mean'], r['median'], r['numgood']] == list(g[tag + '_masked_stats'])
        rn = 1.37 * apref.image_difference(b1, b2, True)['stddev'] / math.sqrt(2)
        rn2 = 2.0 * apref.image_difference(b1, b2, False)['stddev'] / math.sqrt(2)
        assert [rn, rn2] == list(g[tag + '_readnoise'])


def test_g9_bayer_stamps(golden_dir):
    """A9: the reference's own known-answer vectors for RawConv.split (test_core.py:47-259, 14x14 stamps of an
    RGGB mosaic starting at an even row / even column, black level 256): the raw stamp is the sum of the four
    zero-filled channel planes; the split must give back every plane, with and without black subtraction."""
    g = load(golden_dir, 'g9_bayer_stamps.npz')
    chans = ('R', 'G1', 'B', 'G2')                          # plane order of bayer_split
    raw = sum(g[c + '_noblack'] for c in chans).astype(np.uint16)
    assert raw.shape == (14, 14) and (raw > 0).all()
    planes = apref.bayer_split(raw, pattern=(0, 1, 3, 2))
    for k, c in enumerate(chans):
        assert np.array_equal(planes[k], g[c + '_noblack']), c
        mean, std, mn, mx, total = g[c + '_noblack_stats']
        assert planes[k].sum() == total and planes[k].max() == mx and planes[k].min() == mn
    planes = apref.bayer_split(raw, pattern=(0, 1, 3, 2), black=(256, 256, 256, 256))
    for k, c in enumerate(chans):
        assert np.array_equal(planes[k], g[c + '_black']), c
        assert planes[k].sum() == g[c + '_black_stats'][4]


def test_numpy_baseline_restatement_matches_the_oracle():
    """oracle/numpy_ref.py (the NumPy CPU baseline bench.py times) computes what the pinned C oracle computes."""
    from oracle import apref, numpy_ref
    rng = np.random.default_rng(3)
    raw, bias, dark, nflat = numpy_ref.synth_block(16, 6, 40, 9)
    raw[3, 2, 5] = np.nan
    nflat[1, 1] = 0.0
    cal = np.stack([numpy_ref.calibrate_numpy(raw[f], bias, dark, nflat, 0.4) for f in range(raw.shape[0])])
    assert np.array_equal(cal, apref.calibrate(raw, bias, dark, nflat, 0.4), equal_nan=True)
    mean, cnt = numpy_ref.sigclip_numpy(cal, 3.0, 5)
    ref = apref.stack_sigclip(cal, sigma=3.0, maxiters=5, want=('mean', 'count'))
    assert np.array_equal(cnt, ref['count'])
    np.testing.assert_allclose(mean, ref['mean'], rtol=1e-13)
    m2, c2 = numpy_ref.calibrate_stack_numpy(raw, bias, dark, nflat, 0.4)
    assert np.array_equal(c2, cnt) and np.array_equal(m2, mean)


def _g11_case(g, ci):
    m = json.loads(str(g[f'f{ci}_meta']))
    raw = g['raw_' + m['raw']]
    bias, dark = g[m['bias']], g[m['dark']]
    flat = g[m['flat']] if m['flat'] else None
    return m, raw, bias, dark, flat


def test_g11_calibrate_float64(golden_dir):
    """float64 masters / raw frames: NumPy's per-operation promotion, float64 flat normalisation and bad-pixel repair."""
    from oracle import apref
    g = load(golden_dir, 'g11_calibrate_f64.npz')
    for j in range(int(g['nsums'])):
        a = g[f's{j}_a']
        assert apref.lib().apref_pairwise_sum_f64(a.ctypes.data_as(__import__('ctypes').c_void_p), __import__('ctypes').c_long(a.size)) \
            == float(g[f's{j}_sum']), j
        b = a.copy()
        b[::97] = np.nan
        _, norm = apref.flat_normalize(b)
        assert norm == float(g[f's{j}_nanmean']), j
    for ci in range(int(g['ncases'])):
        m, raw, bias, dark, flat = _g11_case(g, ci)
        nflat = None
        if flat is not None:
            nflat, _ = apref.flat_normalize(flat)
            assert nflat.dtype == g[f'f{ci}_nflat'].dtype and np.array_equal(nflat, g[f'f{ci}_nflat'], equal_nan=True), ci
        out = apref.calibrate_mixed(raw, bias, dark, nflat, m['img_exp'] / m['dark_exp'], m['pedestal'], m['dark_still_biased'])
        if m['use_mask']:
            out, st = apref.fix_badpix(out, g['mask'], m['deltapix'])
        ref = g[f'f{ci}_out']
        assert out.dtype == ref.dtype == np.float64 and m['bitpix'] == -64, ci
        assert np.array_equal(out, ref, equal_nan=True), (ci, m)
    # with nothing float64 the mixed routine is the float32 routine
    g1 = load(golden_dir, 'g1_calibrate.npz')
    raw, bias, dark, flat = g1['raw_f32_64x64'], g1['bias_64x64'], g1['dark_64x64'], g1['flat_64x64']
    nf, _ = apref.flat_normalize(flat)
    a = apref.calibrate_mixed(raw, bias, dark, nf, 0.4, -100.0, True)
    assert a.dtype == np.float32 and np.array_equal(a, apref.calibrate(raw, bias, dark, nf, 0.4, -100.0, True), equal_nan=True)
