"""GPU parity tests: the HIP kernels (through the C ABI, libapgpu.so) against the CPU oracle and the
golden vectors captured from the reference.  Run with `pytest -m gpu` on an MI355X."""
import json

import numpy as np
import pytest

from tests.util import (assert_biteq, assert_ulp, load_golden, meta, synth_cube, synth_masters)

pytestmark = pytest.mark.gpu

torch = pytest.importorskip('torch')


@pytest.fixture(scope='module')
def ops():
    if not torch.cuda.is_available():
        pytest.skip('no GPU')
    from astrophotography_amd import ops as _ops
    return _ops


@pytest.fixture(scope='module')
def apref():
    from oracle import apref as _a
    return _a


def dev(a, ops):
    a = np.ascontiguousarray(a)
    if a.dtype == np.uint16:
        return ops.to_device_u16(a)
    return torch.from_numpy(a).cuda()


def host(t):
    if t.dtype == torch.uint16:
        return t.view(torch.int16).cpu().numpy().view(np.uint16)
    return t.cpu().numpy()


# ---- A1 -----------------------------------------------------------------------------------------------
def test_flat_normalize_golden(ops):
    g = load_golden('g7_nanmean.npz')
    for ci in range(int(g['ncases'])):
        a = g[f'a{ci}']
        nflat, norm = ops.flat_normalize(dev(a.reshape(-1), ops))
        assert_biteq(host(norm)[0:1], np.array([g[f'nanmean{ci}']], np.float32), f'norm case {ci}')
        assert_biteq(host(nflat), (a.reshape(-1) / g[f'nanmean{ci}']).astype(np.float32), f'nflat case {ci}')
        b = a.copy().reshape(-1)
        b[::17] = np.nan
        _, normb = ops.flat_normalize(dev(b, ops))
        assert_biteq(host(normb)[0:1], np.array([g[f'nanmean_withnan{ci}']], np.float32), f'norm-with-nan case {ci}')


def test_flat_normalize_large(ops, apref):
    rng = np.random.default_rng(1)
    for n in (8192 * 3, 8192 * 5 + 4097, 1024 * 1024 + 13):
        a = rng.normal(30000, 300, n).astype(np.float32)
        a[rng.integers(0, n, 5)] = np.nan
        nflat, norm = ops.flat_normalize(dev(a, ops))
        rn, rnorm = apref.flat_normalize(a)
        assert_biteq(host(norm)[0:1], np.array([rnorm], np.float32), f'n={n}')
        assert_biteq(host(nflat), rn, f'n={n}')


# ---- A2 (+A5) golden ------------------------------------------------------------------------------------
def test_calibrate_golden(ops):
    g = load_golden('g1_calibrate.npz')
    n = int(g['ncases'])
    for ci in range(n):
        m = meta(g, f'c{ci}_meta')
        H, W = m['shape']
        shp = f'_{H}x{W}'
        raw = dev(g['raw_' + m['raw'] + shp], ops)
        bias, dark = dev(g['bias' + shp], ops), dev(g['dark' + shp], ops)
        nflat = None
        if m['flatmode'] != 'noflat':
            nflat, _ = ops.flat_normalize(dev(g[m['flatmode'] + shp], ops))
            assert_biteq(host(nflat), g['n' + m['flatmode'] + shp], f'nflat case {ci}')
        out = ops.calibrate(raw, bias, dark, nflat, m['img_exp'] / m['dark_exp'], pedestal=m['pedestal'],
                            dark_still_biased=m['dark_still_biased'])
        if m['use_mask']:
            out, st = ops.fix_badpix(out, dev(g['mask' + shp], ops), m['deltapix'])
            hdr = {k: v for k, v, _ in json.loads(str(g[f'c{ci}_hdr']))}
            assert [int(x) for x in host(st)] == [int(hdr['BPIXNBAD']), int(hdr['BPIXNFIX']), int(hdr['BPIXNREM'])]
        assert_biteq(host(out), g[f'c{ci}_out'].astype(np.float32), f'calibrate case {ci} {m}')


def test_calibrate_slab_vs_oracle(ops, apref):
    rng = np.random.default_rng(2)
    for (N, shape, dt) in [(5, (33, 64), np.float32), (4, (16, 37), np.uint16), (3, (7, 9), np.float32), (1, (5, 3), np.uint16)]:
        bias, dark, flat = synth_masters(rng, shape)
        flat[1, 2] = 0.0
        flat[3, 1] = np.nan
        raw = synth_cube(rng, N, shape, dtype=dt)
        e = rng.uniform(0.2, 2.0, N)
        ped = np.where(rng.random(N) < 0.5, 0.0, -100.0)
        nflat_ref, _ = apref.flat_normalize(flat)
        for sb in (False, True):
            for use_flat in (True, False):
                ref = apref.calibrate(raw, bias, dark, nflat_ref if use_flat else None, e, ped, sb)
                out = ops.calibrate(dev(raw, ops), dev(bias, ops), dev(dark, ops), dev(nflat_ref, ops) if use_flat else None,
                                    e, ped, sb)
                assert_biteq(host(out), ref, f'N={N} {shape} {dt} sb={sb} flat={use_flat}')


# ---- A7 golden ------------------------------------------------------------------------------------------
def test_stack_sigclip_golden(ops):
    g = load_golden('g5_stack.npz')
    exact = []
    for ci in range(int(g['ncfg'])):
        cfg = meta(g, f's{ci}_cfg')
        cube = g[f'cube_N{cfg["N"]}']
        r = ops.stack_sigclip(dev(cube, ops), sigma=cfg['sigma'], maxiters=cfg['maxiters'], cenfunc=cfg['cenfunc'],
                              stdfunc=cfg['stdfunc'], outputs=('mean', 'median', 'std', 'count'))
        refmask = np.unpackbits(g[f's{ci}_mask'])[:cube.size].reshape(cube.shape).astype(bool)
        assert np.array_equal(host(r['count']), (~refmask).sum(0)), cfg
        exact.append(assert_ulp(host(r['mean']), g[f's{ci}_mean'].astype(np.float32), 1, f'mean {cfg}'))
        assert_ulp(host(r['median']), g[f's{ci}_median'].astype(np.float32), 1, f'median {cfg}')
        # std: sqrt of a float64 variance computed from shifted moments; 2 ulp(f32) bound
        assert_ulp(host(r['std']), g[f's{ci}_std'].astype(np.float32), 2, f'std {cfg}')
    assert len(exact) == 80 and min(exact) > 0.99
    cfg = meta(g, 'asym_cfg')
    r = ops.stack_sigclip(dev(g['cube_N16'], ops), sigma_lower=cfg['sigma_lower'], sigma_upper=cfg['sigma_upper'],
                          maxiters=cfg['maxiters'])
    assert_ulp(host(r['mean']), g['asym_mean'].astype(np.float32), 1, 'asym')
    r = ops.stack_sigclip(dev(g['u16_cube'], ops), sigma=3.0, maxiters=5, outputs=('mean', 'median', 'std'))
    assert_ulp(host(r['mean']), g['u16_mean'].astype(np.float32), 1, 'u16 mean')
    assert_ulp(host(r['median']), g['u16_median'].astype(np.float32), 1, 'u16 median')


@pytest.mark.parametrize('N', [1, 2, 3, 5, 8, 13, 16, 20, 32, 33, 64, 65, 100, 128, 129, 200, 256, 257, 384, 512])
def test_stack_sigclip_vs_oracle(ops, apref, N):
    rng = np.random.default_rng(100 + N)
    shape = (37, 53)
    cube = synth_cube(rng, N, shape, nan_frac=0.01)
    cube[:, 0, 0] = 7.0
    cube[:, 0, 1] = np.nan
    cube[:, 0, 2] = np.arange(N)
    d = dev(cube, ops)
    for (sigma, maxiters, cen) in [(3.0, 5, 'median'), (2.0, None, 'median'), (3.0, 1, 'mean'), (1.5, 5, 'mean'), (0.5, 3, 'mean')]:
        ref = apref.stack_sigclip(cube, sigma=sigma, maxiters=maxiters, cenfunc=cen)
        r = ops.stack_sigclip(d, sigma=sigma, maxiters=maxiters, cenfunc=cen, outputs=('mean', 'median', 'std', 'count', 'moments'))
        what = f'N={N} sigma={sigma} maxiters={maxiters} cen={cen}'
        assert np.array_equal(host(r['count']), ref['count']), what
        assert_ulp(host(r['mean']), ref['mean'].astype(np.float32), 1, 'mean ' + what)
        assert_ulp(host(r['median']), ref['median'].astype(np.float32), 1, 'median ' + what)
        assert_ulp(host(r['std']), ref['std'].astype(np.float32), 2, 'std ' + what)
        mom = host(r['moments'])
        assert np.array_equal(mom[1], ref['count'].astype(np.float32)), what      # planes: sum, count, sumsq
        kept = np.where(ref['keep'], cube.astype(np.float64), 0.0)
        np.testing.assert_allclose(mom[0], kept.sum(0), rtol=3e-7, atol=1e-30)
        np.testing.assert_allclose(mom[2], (kept * kept).sum(0), rtol=3e-7, atol=1e-30)


def test_big_stacks_other_paths(ops, apref):
    """129 .. 512 frames (LDS-resident column): fused calibration, uint16 frames, mad_std, plain median, pixel mask, float64
    planes and moments - against the oracle; more than 512 frames are refused by the kernel and chunked by ApStack."""
    rng = np.random.default_rng(77)
    for N, dt in ((130, np.float32), (256, np.uint16), (300, np.float32)):
        shape = (9, 70)                                      # 630 pixels: a partly filled last workgroup
        bias, dark, flat = synth_masters(rng, shape)
        flat[0, 0] = 0.0
        flat[0, 1] = np.nan
        raw = synth_cube(rng, N, shape, dtype=dt)
        if dt == np.float32:
            raw = raw + bias + 0.4 * dark
            raw[5, 3, 4] = np.inf
        nflat, _ = apref.flat_normalize(flat)
        e = rng.uniform(0.3, 0.5, N)
        ped = np.where(rng.random(N) < 0.3, -50.0, 0.0)
        pm = (rng.random(shape) < 0.05).astype(np.uint8)
        cal = apref.calibrate(raw, bias, dark, nflat, e, ped, True)
        calib = dict(bias=dev(bias, ops), dark=dev(dark, ops), nflat=dev(nflat, ops), exp_ratio=e, pedestal=ped, dark_still_biased=True)
        d = dev(raw, ops)
        for kw in (dict(sigma=3.0, maxiters=5), dict(sigma=5.0, maxiters=1, cenfunc='median', stdfunc='mad_std'),
                   dict(sigma=2.5, maxiters=None, cenfunc='mean')):
            ref = apref.stack_sigclip(cal, pixmask=pm, **kw)
            r = ops.stack_sigclip(d, calib=calib, pixmask=dev(pm, ops), outputs=('mean', 'median', 'std', 'count', 'mean_f64', 'moments_f64'), **kw)
            what = f'big N={N} {dt.__name__} {kw}'
            assert np.array_equal(host(r['count']), ref['count']), what
            assert_ulp(host(r['mean']), ref['mean'].astype(np.float32), 1, 'mean ' + what)
            assert_ulp(host(r['median']), ref['median'].astype(np.float32), 1, 'median ' + what)
            assert_ulp(host(r['std']), ref['std'].astype(np.float32), 2, 'std ' + what)
            np.testing.assert_allclose(host(r['mean_f64']), ref['mean'], rtol=1e-13, equal_nan=True)
            assert np.array_equal(host(r['moments_f64']['count']), ref['count']), what
        med = ops.stack_median(d, calib=calib)
        assert_ulp(host(med), apref.stack_median(cal).astype(np.float32), 1, f'big median N={N}')
    with pytest.raises(Exception):
        ops.stack_sigclip(torch.zeros((513, 4, 64), device='cuda'))
    # beyond 512 frames: ApStack reduces chunk by chunk; an unclipped mean is exact, the clipped mean is hierarchical
    import astrophotography_amd as ap
    cube = synth_cube(rng, 600, (6, 64))
    d = dev(cube, ops)
    st = ap.ApStack('CRITICAL')
    r = st.stack(d, method='mean', outputs=('mean', 'count'))
    np.testing.assert_allclose(host(r['mean']), cube.astype(np.float64).mean(0), rtol=1.2e-7)
    assert int(r['count'].min()) == 600
    r = st.stack(d, method='sigclip', outputs=('mean', 'count', 'std'))
    tot = np.zeros((6, 64))
    cnt = np.zeros((6, 64), np.int64)
    for lo in (0, 300):
        rr = apref.stack_sigclip(cube[lo:lo + 300], sigma=3.0, maxiters=5, want=('keep', 'count'))
        tot += np.where(rr['keep'], cube[lo:lo + 300].astype(np.float64), 0).sum(0)
        cnt += rr['count']
    assert np.array_equal(host(r['count']), cnt)
    assert_ulp(host(r['mean']), (tot / cnt).astype(np.float32), 1, 'chunked clip = oracle per chunk, moments added')


def test_combine_ccdproc_config_golden(ops):
    """A6: the ccdproc.combine configuration of scripts/ap_combine_darks.py:394-420 (one pass, np.ma.median / mad_std,
    strict 5-sigma bounds, masked mean in float64) against golden group G12 - numpy.ma + astropy run on float64 masked
    cubes, what ccdproc's Combiner calls (tests/golden/make_golden_combine.py)."""
    g = load_golden('g12_combine.npz')
    ncase = 0
    for m in json.loads(str(g['_meta'])):
        if m['kind'] in ('f64ties', 'f64bounds'):
            continue                                         # float64 frames: the kernels take uint16 / float32 slabs
        k = m['case']
        fr = g[f'c{k}_frames']
        # both published forms of Combiner.sigma_clipping: the legacy loop (arrays c<k>_*) and astropy.stats.sigma_clip
        # (c<k>_b_*, run for real: a column holding a non-finite value is not clipped - APGPU_STACK_NONFINITE_UNCLIPPED)
        for tag, flag in (('', False), ('b_', True)):
            r = ops.stack_sigclip(dev(fr, ops), sigma=5.0, maxiters=1, cenfunc='median', stdfunc='mad_std',
                                  outputs=('mean', 'count', 'mean_f64', 'std_f64'), nonfinite_unclipped=flag)
            assert np.array_equal(host(r['count']), g[f'c{k}_{tag}count']), (m, tag)
            ref = g[f'c{k}_{tag}mean']
            np.testing.assert_allclose(host(r['mean_f64']), ref, rtol=4e-16, atol=0, equal_nan=True, err_msg=str(m))
            assert_ulp(host(r['mean']), ref.astype(np.float32), 1, str(m))
            np.testing.assert_allclose(host(r['std_f64']), g[f'c{k}_{tag}std'], rtol=1e-12, atol=1e-12, equal_nan=True, err_msg=str(m))
        ncase += 1
    assert ncase == 12


def test_stack_u16_and_pixmask(ops, apref):
    rng = np.random.default_rng(7)
    cube = synth_cube(rng, 16, (19, 40), dtype=np.uint16)
    pm = (rng.random((19, 40)) < 0.1).astype(np.uint8)
    ref = apref.stack_sigclip(cube, sigma=3.0, maxiters=5, pixmask=pm)
    r = ops.stack_sigclip(dev(cube, ops), sigma=3.0, maxiters=5, pixmask=dev(pm, ops), outputs=('mean', 'count'))
    assert np.array_equal(host(r['count']), ref['count'])
    assert_ulp(host(r['mean']), ref['mean'].astype(np.float32), 1, 'u16 pixmask')
    assert np.isnan(host(r['mean'])[pm != 0]).all()


@pytest.mark.parametrize('dt', [np.float32, np.uint16])
def test_fused_calibrate_stack_vs_oracle(ops, apref, dt):
    rng = np.random.default_rng(11)
    for N, shape in [(8, (24, 64)), (64, (16, 48)), (20, (9, 21))]:
        bias, dark, flat = synth_masters(rng, shape)
        flat[0, 0] = 0.0
        flat[0, 1] = np.nan
        raw = synth_cube(rng, N, shape, dtype=dt)
        raw = (raw.astype(np.float64) + 1000).astype(dt) if dt == np.float32 else (raw + 1000).astype(dt)
        e = np.full(N, 120.0 / 300.0)
        e[::3] = 0.5
        nflat, _ = apref.flat_normalize(flat)
        for sb in (False, True):
            ref_mean, ref_cnt = apref.calibrate_stack(raw, bias, dark, nflat, e, None, sb, sigma=3.0, maxiters=5)
            calib = dict(bias=dev(bias, ops), dark=dev(dark, ops), nflat=dev(nflat, ops), exp_ratio=e, dark_still_biased=sb)
            r = ops.stack_sigclip(dev(raw, ops), sigma=3.0, maxiters=5, calib=calib, outputs=('mean', 'count'))
            assert np.array_equal(host(r['count']), ref_cnt), (N, shape, sb)
            assert_ulp(host(r['mean']), ref_mean, 1, f'fused N={N} {shape} sb={sb}')
            # fused == unfused on the GPU: the same survivors; the means are roundings of the same sums by two kernels
            # (float32 sum + one rounding on the fast path, float64 sum on the exact path - which pixel takes which differs)
            cal = ops.calibrate(dev(raw, ops), calib['bias'], calib['dark'], calib['nflat'], e, None, sb)
            r2 = ops.stack_sigclip(cal, sigma=3.0, maxiters=5, outputs=('mean', 'count'))
            assert np.array_equal(host(r2['count']), host(r['count'])), 'fused vs unfused'
            assert_ulp(host(r2['mean']), host(r['mean']), 1, 'fused vs unfused')


def test_stack_median_vs_oracle(ops, apref):
    rng = np.random.default_rng(13)
    for N in (1, 2, 5, 8, 16, 31, 64, 100):
        cube = synth_cube(rng, N, (11, 70), nan_frac=0.02)
        cube[~np.isfinite(cube) & (rng.random(cube.shape) < 0.5)] = np.nan
        ref = apref.stack_median(cube)
        med = ops.stack_median(dev(cube, ops))
        assert_ulp(host(med), ref.astype(np.float32), 0 if N % 2 else 1, f'median N={N}')
    g = load_golden('g6_madstd.npz')
    for N in (5, 8, 16):
        med = ops.stack_median(dev(g[f'cube_N{N}'], ops))
        assert_ulp(host(med), g[f'median_N{N}'].astype(np.float32), 1, f'golden median N={N}')


def test_stack_properties_large(ops):
    """Size-independent properties at a larger size: permutation invariance (bit-exact: the kernel sums
    in sorted order), constant frames, and rejection of a planted outlier."""
    g = torch.Generator(device='cuda').manual_seed(5)
    N, H, W = 64, 512, 1024
    cube = torch.randn((N, H, W), generator=g, device='cuda') * 20 + 500
    cube[7, ::5, ::7] += 4000
    r1 = ops.stack_sigclip(cube, outputs=('mean', 'count'))
    perm = torch.randperm(N, device='cuda')
    r2 = ops.stack_sigclip(cube[perm].contiguous(), outputs=('mean', 'count'))
    assert torch.equal(r1['mean'], r2['mean']) and torch.equal(r1['count'], r2['count'])
    assert int(r1['count'][::5, ::7].max()) <= N - 1
    plain = cube.double().mean(0)
    assert float((r1['mean'].double() - plain).abs().max()) < 100
    const = torch.full((N, 64, 64), 123.25, device='cuda')
    rc = ops.stack_sigclip(const, outputs=('mean', 'std', 'count'))
    assert torch.all(rc['mean'] == 123.25) and torch.all(rc['std'] == 0) and torch.all(rc['count'] == N)
    mean = ops.moments_finalize(ops.stack_sigclip(cube, outputs=('moments',))['moments'])
    assert float((mean - r1['mean']).abs().max()) < 1e-3
    with pytest.raises(ValueError):                       # float32 sums about zero cannot give a std (cancellation)
        ops.moments_finalize(ops.stack_sigclip(cube, outputs=('moments',))['moments'], want_std=True)
    # the float64 layout: mean is the kernel's float64 mean rounded once (bit-equal), std within 2 ulp of the two-pass plane
    rs = ops.stack_sigclip(cube, outputs=('mean', 'std', 'count'))
    m64 = ops.stack_sigclip(cube, outputs=('moments_f64',))['moments_f64']
    assert torch.equal(m64['count'], rs['count'])
    (mean2, std2), (mean64, std64) = ops.moments_finalize(m64, want_f64=True)
    assert_ulp(host(mean2), host(rs['mean']), 1, 'mean from float64 moments')
    assert float(((std2 - rs['std']).abs() / rs['std']).max()) < 1e-5
    assert torch.equal(mean64.float(), mean2)


# ---- A3 / A4 ---------------------------------------------------------------------------------------------
def test_sigclip_global_and_mask_golden(ops):
    g = load_golden('g2_findbadpix.npz')
    for ci in range(int(g['ncases'])):
        dark = g[f'd{ci}_dark']
        d = dev(dark, ops)
        st = ops.sigclip_global(d, sigma=4.0, maxiters=5)
        s = host(st)
        ref = g[f'd{ci}_stats']
        if dark.dtype == np.float32:
            assert_biteq(s[:3].astype(np.float32), ref.astype(np.float32), f'stats case {ci}')
        else:                                   # integer dark: numpy's float64 statistics, reproduced exactly
            assert list(s[:3]) == list(ref), ci
            d = dev(dark.astype(np.float32), ops)
        lo = float(s[1]) - 4.0 * float(s[2])
        hi = float(s[1]) + 4.0 * float(s[2])
        assert [lo, hi] == list(g[f'd{ci}_thresh'])
        mask, nbad = ops.threshold_mask(d, lo, hi)
        assert np.array_equal(host(mask), g[f'd{ci}_mask_auto'])
        assert int(host(nbad)[0]) == int(g[f'd{ci}_nbad_auto'])
        thr = torch.tensor([lo, hi], dtype=torch.float64, device='cuda')
        mask2, _ = ops.threshold_mask(d, thresholds=thr)
        assert torch.equal(mask, mask2)
        if f'd{ci}_mask_user' in g:
            H, W = dark.shape
            rects = [[0, H, c - 1, c] for c in (12, 13, 17)] + [[0, 1, 0, 1], [4, 6, 6, 12], [199, 300, 399, 420]]
            ops.mask_add_rects(mask, rects, 2)
            assert np.array_equal(host(mask), g[f'd{ci}_mask_user'])


def test_sigclip_global_vs_oracle(ops, apref):
    rng = np.random.default_rng(17)
    for n in (1, 7, 100, 8191, 8192, 8193, 70001, 1 << 20):
        x = rng.normal(20, 3, n).astype(np.float32)
        hot = rng.random(n) < 0.001
        x[hot] = rng.uniform(2000, 6000, hot.sum())
        if n > 50:
            x[rng.integers(0, n, 3)] = np.nan
            x[rng.integers(0, n, 2)] = np.inf
        # (1.5, None) and (1.0, 40): far more passes than the 32 the entry point queues without looking at the convergence flag
        for sigma, maxiters in [(4.0, 5), (3.0, 2), (2.5, None), (1.5, None), (1.0, 40)]:
            ref = apref.sigclip_global(x, sigma=sigma, maxiters=maxiters)
            s = host(ops.sigclip_global(dev(x, ops), sigma=sigma, maxiters=maxiters))
            what = f'n={n} sigma={sigma} maxiters={maxiters}'
            assert_biteq(s[:3].astype(np.float32), np.array([ref['mean'], ref['median'], ref['std']], np.float32), what)
            assert int(s[6]) == ref['nkeep'] and int(s[5]) == ref['niter'], what
            assert (s[3] == ref['lo'] and s[4] == ref['hi']) or (np.isnan(s[3]) and np.isnan(ref['lo'])), what


# ---- A5 --------------------------------------------------------------------------------------------------
def test_fix_badpix_golden(ops):
    g = load_golden('g3_fixbadpix.npz')
    data, mask = dev(g['data'], ops), dev(g['mask'], ops)
    for dp in (1, 2, 3):
        out, st = ops.fix_badpix(data, mask, dp)
        assert_biteq(host(out), g[f'out_dp{dp}'], f'dp={dp}')
        ref = json.loads(str(g[f'stats_dp{dp}']))
        assert [int(x) for x in host(st)] == [ref['BPIXNBAD'][0], ref['BPIXNFIX'][0], ref['BPIXNREM'][0]]
    out, st = ops.fix_badpix(data, torch.zeros_like(mask), 1)
    assert_biteq(host(out), g['out_emptymask'])
    assert [int(x) for x in host(st)] == [0, 0, 0]
    with pytest.raises(RuntimeError):
        ops.fix_badpix(data, mask[:10], 1)


def test_fix_badpix_vs_oracle(ops, apref):
    rng = np.random.default_rng(19)
    data = rng.normal(500, 20, (301, 257)).astype(np.float32)
    data[5, 5] = np.nan
    mask = (rng.random(data.shape) < 0.03).astype(np.uint8) * 3
    mask[100:110, 50:60] = 1
    for dp in (1, 2):
        ref, st = apref.fix_badpix(data, mask, dp)
        out, s = ops.fix_badpix(dev(data, ops), dev(mask, ops), dp)
        assert_biteq(host(out), ref)
        assert [int(x) for x in host(s)] == [st['nbad'], st['nfix'], st['nrem']]


# ---- A8 / A9 ----------------------------------------------------------------------------------------------
def test_imarith_golden(ops):
    g = load_golden('g4_imarith.npz')
    a, b, au, bu = (dev(g[k], ops) for k in ('a', 'b', 'au', 'bu'))
    for op in ('ADD', 'SUB', 'MUL', 'DIV'):
        assert_biteq(host(ops.imarith(a, op, b)), g[f'f32_arr_{op}'].astype(np.float32), op)
        assert_biteq(host(ops.imarith(a, op, 3.25)), g[f'f32_scl_{op}'].astype(np.float32), op)
    for op in ('ADD', 'SUB', 'MUL'):
        assert np.array_equal(host(ops.imarith(au, op, bu)), g[f'u16_arr_{op}'])
    from astrophotography_amd._lib import ApGpuError
    with pytest.raises(ApGpuError):
        ops.imarith(au, 'DIV', bu)
    with pytest.raises(ApGpuError):
        ops.imarith(au, 'ADD', 2.0)


def test_bayer_split_vs_oracle(ops, apref):
    rng = np.random.default_rng(23)
    raw = rng.integers(0, 16384, (46, 62)).astype(np.uint16)
    for pattern, black in [((0, 1, 3, 2), None), ((0, 1, 3, 2), (256, 256, 256, 256)), ((2, 3, 1, 0), (100, 0, 50, 7))]:
        ref = apref.bayer_split(raw, pattern, black)
        out = ops.bayer_split(dev(raw, ops), pattern, black)
        assert np.array_equal(host(out), ref)


def test_bayer_split_reference_stamps(ops):
    """The reference's RawConv.split known-answer stamps (golden G9) through the HIP kernel."""
    g = load_golden('g9_bayer_stamps.npz')
    chans = ('R', 'G1', 'B', 'G2')
    raw = sum(g[c + '_noblack'] for c in chans).astype(np.uint16)
    out = host(ops.bayer_split(dev(raw, ops), (0, 1, 3, 2)))
    outb = host(ops.bayer_split(dev(raw, ops), (0, 1, 3, 2), (256, 256, 256, 256)))
    for k, c in enumerate(chans):
        assert np.array_equal(out[k], g[c + '_noblack']) and np.array_equal(outb[k], g[c + '_black']), c


def test_errors_are_loud(ops):
    from astrophotography_amd._lib import ApGpuError
    with pytest.raises(ApGpuError):
        ops.stack_sigclip(torch.zeros((513, 4, 4), device='cuda'))              # beyond APGPU_MAX_STACK
    with pytest.raises(ValueError):
        ops.stack_sigclip(torch.zeros((4, 4, 4)))


def test_fast_division_is_ieee_exact(ops):
    """The fused kernel divides by the per-pixel flat through a precomputed reciprocal + two FMA
    corrections; a single-frame stack returns the calibrated value itself, so it must equal the
    unfused calibrate kernel (IEEE division) bit for bit - 2 x 16M random quotients plus hard cases."""
    g = torch.Generator(device='cuda').manual_seed(31)
    H, W = 4096, 4096
    for trial in range(2):
        mag = torch.rand((H, W), generator=g, device='cuda') * 60 - 30            # 2^-30 .. 2^30
        raw = torch.exp2(mag) * (torch.rand((H, W), generator=g, device='cuda') + 1)
        raw = torch.where(torch.rand((H, W), generator=g, device='cuda') < 0.5, raw, -raw)
        nmag = torch.rand((H, W), generator=g, device='cuda') * 20 - 10
        nflat = torch.exp2(nmag) * (torch.rand((H, W), generator=g, device='cuda') + 1)
        if trial == 1:
            # divisors with extreme significands (all ones / one above a power of two) and tiny quotient gaps
            bits = torch.randint(0, 1 << 23, (H, W), generator=g, device='cuda', dtype=torch.int32)
            special = torch.tensor([0x7FFFFF, 0x000001, 0x7FFFFE, 0x400000, 0x3FFFFF], device='cuda', dtype=torch.int32)
            pick = torch.randint(0, 5, (H, W), generator=g, device='cuda')
            bits = torch.where(torch.rand((H, W), generator=g, device='cuda') < 0.5, special[pick], bits)
            nflat = (bits | (127 << 23)).view(torch.float32)
            raw = (torch.randint(0, 1 << 23, (H, W), generator=g, device='cuda', dtype=torch.int32) | (130 << 23)).view(torch.float32)
        bias = torch.zeros((H, W), device='cuda')
        dark = torch.zeros((H, W), device='cuda')
        calib = dict(bias=bias, dark=dark, nflat=nflat, exp_ratio=1.0)
        exact = ops.calibrate(raw, bias, dark, nflat, 1.0)
        fused = ops.stack_sigclip(raw[None], calib=calib, outputs=('mean', 'count'))
        assert int(fused['count'].min()) == 1
        assert torch.equal(exact.view(torch.int32), fused['mean'].view(torch.int32)), trial
    # out-of-range operands must take the exact fallback, not the fast path
    raw = torch.full((64, 64), 3.0e-39, device='cuda')                      # subnormal numerator
    nflat = torch.full((64, 64), 3.0, device='cuda')
    z = torch.zeros((64, 64), device='cuda')
    exact = ops.calibrate(raw, z, z, nflat, 1.0)
    fused = ops.stack_sigclip(raw[None], calib=dict(bias=z, dark=z, nflat=nflat, exp_ratio=1.0), outputs=('mean',))
    assert torch.equal(exact.view(torch.int32), fused['mean'].view(torch.int32))


def test_full_size_c2_matches_oracle(ops, apref):
    """BASELINE.json configs[1] at full size (64 x 4096 x 4096 f32, fused calibrate + 3-sigma clip): every
    one of the 16.7 M output pixels against the CPU oracle - identical survivor counts, mean within 1 ulp."""
    from astrophotography_amd import synth
    N, H, W = 64, 4096, 4096
    masters = synth.make_masters(H, W, config_id=2, device='cuda')
    nflat, _ = ops.flat_normalize(masters['flat'])
    frames = synth.make_frames(N, masters, nflat, config_id=2)
    calib = dict(bias=masters['bias'], dark=masters['dark'], nflat=nflat, exp_ratio=synth.EXP_RATIO)
    r = ops.stack_sigclip(frames, sigma=3.0, maxiters=5, calib=calib, outputs=('mean', 'count'))
    torch.cuda.synchronize()
    mean, cnt = host(r['mean']), host(r['count'])
    ref_mean = np.empty((H, W), np.float32)
    ref_cnt = np.empty((H, W), np.int32)
    b, d, nf = host(masters['bias']), host(masters['dark']), host(nflat)
    for r0 in range(0, H, 1024):                         # 1 GiB of frames at a time through host memory
        sl = slice(r0, r0 + 1024)
        m, c = apref.calibrate_stack(host(frames[:, sl]), b[sl], d[sl], nf[sl], synth.EXP_RATIO, sigma=3.0, maxiters=5)
        ref_mean[sl], ref_cnt[sl] = m, c
    assert np.array_equal(cnt, ref_cnt)
    exact = assert_ulp(mean, ref_mean, 1, 'C2 full size')
    assert exact > 0.995          # float32 fast path: the sum of the deviations carries ~2e-3 ulp of float32 rounding (99.8 % measured)
    assert 40 <= cnt.min() and cnt.max() == 64 and (cnt < 64).mean() > 0.05      # the clip did real work


def test_config4_bayer_u16_median_pipeline(ops, apref):
    """BASELINE configs[3] at reduced size: uint16 Bayer mosaic frames, per-channel flat normalisation,
    fused calibration + median stack, then the R/G1/B/G2 split geometry on the stacked result."""
    rng = np.random.default_rng(41)
    N, H, W = 64, 40, 96
    gains = np.array([[0.55, 1.0], [1.0, 0.7]], np.float32)              # R G1 / G2 B
    gain_map = np.tile(gains, (H // 2, W // 2))
    bias = rng.normal(1000, 5, (H, W)).astype(np.float32)
    dark = rng.normal(20, 3, (H, W)).astype(np.float32)
    flat = (rng.normal(30000, 300, (H, W)) * gain_map).astype(np.float32)
    sky = 2000 * gain_map
    raw = np.clip(np.rint(bias + 0.4 * dark + sky + rng.normal(0, 30, (N, H, W))), 0, 65535).astype(np.uint16)
    raw[rng.random(raw.shape) < 0.002] = 60000
    nflat, norms = ops.bayer_flat_normalize(dev(flat, ops))
    ref_nflat = np.empty_like(flat)
    for r0 in (0, 1):
        for c0 in (0, 1):
            pl, nm = apref.flat_normalize(np.ascontiguousarray(flat[r0::2, c0::2]))
            ref_nflat[r0::2, c0::2] = pl
            assert host(norms)[r0, c0] == nm
    assert_biteq(host(nflat), ref_nflat, 'per-channel nflat')
    calib = dict(bias=dev(bias, ops), dark=dev(dark, ops), nflat=nflat, exp_ratio=0.4)
    med = ops.stack_median(dev(raw, ops), calib=calib)
    ref = apref.stack_median(apref.calibrate(raw, bias, dark, ref_nflat, 0.4))
    assert_ulp(host(med), ref.astype(np.float32), 1, 'C4 median stack')
    # channel geometry (RawConv.split): RGGB, G1 on the red row
    planes = ops.bayer_split(dev(raw[0], ops), (0, 1, 3, 2), (256, 256, 256, 256))
    assert np.array_equal(host(planes), apref.bayer_split(raw[0], (0, 1, 3, 2), (256, 256, 256, 256)))
    p = host(planes)
    assert (p[0][1::2] == 0).all() and (p[0][:, 1::2] == 0).all() and (p[2][::2] == 0).all()


def test_sigclip_global_f64_vs_oracle(ops, apref):
    """Integer / float64 images: numpy runs every statistic in float64; same tree, same bits."""
    rng = np.random.default_rng(43)
    for n in (5, 1000, 8192 * 2 + 77, 300000):
        xu = np.clip(np.rint(rng.normal(1000, 8, n)), 0, 65535).astype(np.uint16)
        xu[rng.random(n) < 0.002] = 5000
        xd = rng.normal(0, 3, n)
        xd[rng.random(n) < 0.002] += 100
        if n > 100:
            xd[3] = np.nan
        for x in (xu, xd):
            for sigma, maxiters in ((3.0, 5), (4.0, None), (1e300, 1)):
                ref = apref.sigclip_global(x, sigma=sigma, maxiters=maxiters)
                t = ops.to_device_u16(x) if x.dtype == np.uint16 else torch.from_numpy(x).cuda()
                s = host(ops.sigclip_global(t, sigma=sigma, maxiters=maxiters))
                what = f'n={n} {x.dtype} sigma={sigma} maxiters={maxiters}'
                assert [s[0], s[1], s[2]] == [ref['mean'], ref['median'], ref['std']], what
                assert int(s[6]) == ref['nkeep'] and int(s[5]) == ref['niter'], what
                fin = x[np.isfinite(x.astype(np.float64))].astype(np.float64)
                if sigma > 1e100:
                    assert s[7] == fin.min() and s[8] == fin.max(), what


def test_stack_median_u16_pairs_paths(ops, apref):
    """uint16 median stacks: the packed two-pixels-per-lane kernel (raw columns sorted, only the two middle
    values calibrated - valid because a uniform calibration is monotone) against 'oracle calibrate, then
    oracle median', with every special value of the masters; and the cases that must take the ordinary path
    (per-frame exposure ratios, pedestals, odd pixel counts)."""
    rng = np.random.default_rng(77)
    shape = (12, 86)                                        # P = 1032: even -> pairs
    bias, dark, flat = synth_masters(rng, shape)
    nflat = (flat / np.float32(30000.0)).astype(np.float32)
    nflat[0, :8] = [0.0, np.nan, -1.25, np.inf, -np.inf, 1e-30, -0.0, 2.0]
    bias[1, :3] = [np.nan, np.inf, -np.inf]
    dark[2, :3] = [np.nan, np.inf, -np.inf]
    pixmask = (rng.random(shape) < 0.01).astype(np.uint8)
    for N in (1, 2, 5, 16, 37, 64, 100):
        cube = synth_cube(rng, N, shape, dtype=np.uint16)
        cube[:, 3, :4] = [0, 65535, 1, 65534]               # extremes; identical columns
        for sb in (False, True):
            for e_kind in ('uniform', 'perframe', 'pedestal'):
                e = np.full(N, 0.4, np.float32)
                ped = None
                if e_kind == 'perframe' and N > 1:
                    e = rng.uniform(0.2, 0.6, N).astype(np.float32)
                if e_kind == 'pedestal':
                    ped = np.zeros(N, np.float32)
                    ped[N // 2] = -100.0
                cal = apref.calibrate(cube, bias, dark, nflat, e, pedestal=ped, dark_still_biased=sb)
                ref = apref.stack_median(cal)
                nref = (~np.isnan(cal)).sum(0).astype(np.int32)
                ref[pixmask != 0] = np.nan
                nref[pixmask != 0] = 0
                calib = dict(bias=dev(bias, ops), dark=dev(dark, ops), nflat=dev(nflat, ops), exp_ratio=dev(e, ops),
                             pedestal=None if ped is None else dev(ped, ops), dark_still_biased=sb)
                med, cnt = ops.stack_median(dev(cube, ops), calib=calib, pixmask=dev(pixmask, ops), want_count=True)
                assert np.array_equal(host(cnt), nref), (N, sb, e_kind)
                assert_ulp(host(med), ref.astype(np.float32), 0 if N % 2 else 1, f'u16 median N={N} sb={sb} {e_kind}')
        # plain (no calibration) and an odd pixel count (ordinary kernel)
        med = ops.stack_median(dev(cube, ops))
        assert_ulp(host(med), apref.stack_median(cube.astype(np.float32)).astype(np.float32), 0, f'plain u16 median N={N}')
        odd = np.ascontiguousarray(cube[:, :, :85])
        calib = dict(bias=dev(np.ascontiguousarray(bias[:, :85]), ops), dark=dev(np.ascontiguousarray(dark[:, :85]), ops),
                     nflat=dev(np.ascontiguousarray(nflat[:, :85]), ops), exp_ratio=0.4)
        cal = apref.calibrate(odd, bias[:, :85], dark[:, :85], nflat[:, :85], np.full(N, 0.4, np.float32))
        assert_ulp(host(ops.stack_median(dev(odd, ops), calib=calib)), apref.stack_median(cal).astype(np.float32),
                   0 if N % 2 else 1, f'odd-P u16 median N={N}')


def test_stack_sigclip_u16_pairs_paths(ops, apref):
    """uint16 clipped stacks through the two-pixels-per-lane kernel (raw columns sorted with packed 16-bit
    compare-exchanges, then calibrated): counts equal to / means within 1 ulp of 'oracle calibrate, oracle clip',
    and bit-identical to the one-pixel-per-lane kernel fed the same values as float32 - for every special value
    of the masters (flat 0 / NaN / negative / inf, non-finite bias and dark), pixel masks, padded stacks, and the
    cases that must take the ordinary path (per-frame exposure ratios, pedestals, odd pixel counts)."""
    rng = np.random.default_rng(78)
    shape = (10, 90)                                        # P = 900: even -> pairs
    bias, dark, flat = synth_masters(rng, shape)
    nflat = (flat / np.float32(30000.0)).astype(np.float32)
    nflat[0, :8] = [0.0, np.nan, -1.25, np.inf, -np.inf, 1e-30, -0.0, 2.0]
    bias[1, :3] = [np.nan, np.inf, -np.inf]
    dark[2, :3] = [np.nan, np.inf, -np.inf]
    pixmask = (rng.random(shape) < 0.01).astype(np.uint8)
    for N in (1, 3, 8, 16, 37, 64, 100):
        cube = synth_cube(rng, N, shape, dtype=np.uint16)
        cube[:, 3, :4] = [0, 65535, 1, 65534]
        for sb, e_kind in ((False, 'uniform'), (True, 'uniform'), (False, 'perframe'), (False, 'pedestal')):
            e = np.full(N, 0.4, np.float32)
            ped = None
            if e_kind == 'perframe' and N > 1:
                e = rng.uniform(0.2, 0.6, N).astype(np.float32)
            if e_kind == 'pedestal':
                ped = np.zeros(N, np.float32)
                ped[N // 2] = -100.0
            cal = apref.calibrate(cube, bias, dark, nflat, e, pedestal=ped, dark_still_biased=sb)
            with np.errstate(all='ignore'):
                ref = apref.stack_sigclip(cal, sigma=3.0, maxiters=5)
            mref, nref = ref['mean'].astype(np.float32), ref['count'].copy()
            mref[pixmask != 0] = np.nan
            nref[pixmask != 0] = 0
            calib = dict(bias=dev(bias, ops), dark=dev(dark, ops), nflat=dev(nflat, ops), exp_ratio=dev(e, ops),
                         pedestal=None if ped is None else dev(ped, ops), dark_still_biased=sb)
            what = f'u16 pairs N={N} sb={sb} {e_kind}'
            r = ops.stack_sigclip(dev(cube, ops), sigma=3.0, maxiters=5, calib=calib, pixmask=dev(pixmask, ops),
                                  outputs=('mean', 'count', 'moments'))
            assert np.array_equal(host(r['count']), nref), what
            assert_ulp(host(r['mean']), mref, 1, what)
            # on the float64 path (APGPU_STACK_EXACT_MOMENTS) both kernels sum the survivors in sorted order: bit-identical;
            # the default float32 fast path sums a pruned-sorted core, so the two agree to float32 rounding only
            rx = ops.stack_sigclip(dev(cube, ops), sigma=3.0, maxiters=5, calib=calib, pixmask=dev(pixmask, ops),
                                   outputs=('mean', 'count', 'moments'), exact=True)
            rf = ops.stack_sigclip(dev(cube.astype(np.float32), ops), sigma=3.0, maxiters=5, calib=calib,
                                   pixmask=dev(pixmask, ops), outputs=('mean', 'count', 'moments'), exact=True)
            assert_biteq(host(rx['mean']), host(rf['mean']), what + ' vs float32 kernel')
            assert_biteq(host(rx['moments']), host(rf['moments']), what + ' moments vs float32 kernel')
            assert np.array_equal(host(rx['count']), nref), what
            assert_ulp(host(rx['mean']), host(r['mean']), 1, what + ' float32 fast path vs float64 path')
            np.testing.assert_allclose(host(r['moments'])[0], host(rx['moments'])[0], rtol=1e-6, equal_nan=True)
        # plain (no calibration), mean-centred, and an odd pixel count (ordinary kernel)
        with np.errstate(all='ignore'):
            ref = apref.stack_sigclip(cube.astype(np.float32), sigma=2.5, maxiters=None, cenfunc='mean')
        r = ops.stack_sigclip(dev(cube, ops), sigma=2.5, maxiters=None, cenfunc='mean', outputs=('mean', 'count'))
        assert np.array_equal(host(r['count']), ref['count']) and True
        assert_ulp(host(r['mean']), ref['mean'].astype(np.float32), 1, f'plain u16 pairs N={N}')
        odd = np.ascontiguousarray(cube[:, :, :89])
        ro = ops.stack_sigclip(dev(odd, ops), sigma=3.0, maxiters=5, outputs=('mean',))
        rfo = ops.stack_sigclip(dev(odd.astype(np.float32), ops), sigma=3.0, maxiters=5, outputs=('mean',))
        assert_biteq(host(ro['mean']), host(rfo['mean']), f'odd-P u16 N={N}')


@pytest.mark.parametrize('N', [9, 12, 14, 20, 24, 30, 36, 40, 45, 48, 52, 56, 61, 70, 77, 80, 93, 96, 100, 112])
def test_stack_three_quarter_slot_sizes(ops, apref, N):
    """Slot counts 12 / 24 / 40 / 48 / 56 / 80 / 96 / 112 use Batcher networks pruned to their first NP wires and non-power-of-two
    multiplexer trees: full (N = NP) and padded stacks, lean and rich outputs, median, float32 and uint16."""
    rng = np.random.default_rng(900 + N)
    shape = (6, 128)
    cube = synth_cube(rng, N, shape, nan_frac=0.01)
    bias, dark, flat = synth_masters(rng, shape)
    nflat = (flat / np.float32(30000.0)).astype(np.float32)
    e = np.full(N, 0.4, np.float32)
    cal = apref.calibrate(cube, bias, dark, nflat, e)
    calib = dict(bias=dev(bias, ops), dark=dev(dark, ops), nflat=dev(nflat, ops), exp_ratio=dev(e, ops))
    for cen, dv, sig, mi in (('median', 'std', 3.0, 5), ('mean', 'std', 2.5, None), ('median', 'mad_std', 5.0, 1)):
        with np.errstate(all='ignore'):
            ref = apref.stack_sigclip(cal, sigma=sig, maxiters=mi, cenfunc=cen, stdfunc=dv)
        what = f'N={N} {cen}/{dv}'
        r = ops.stack_sigclip(dev(cube, ops), sigma=sig, maxiters=mi, cenfunc=cen, stdfunc=dv, calib=calib,
                              outputs=('mean', 'count') if dv == 'std' else ('mean', 'count', 'median', 'std'))
        assert np.array_equal(host(r['count']), ref['count']), what
        assert_ulp(host(r['mean']), ref['mean'].astype(np.float32), 1, what)
        if 'median' in r:
            assert_ulp(host(r['median']), ref['median'].astype(np.float32), 1, what + ' median plane')
            assert_ulp(host(r['std']), ref['std'].astype(np.float32), 2, what + ' std plane')
        elif cen == 'median':
            # the three planes of a padded stack on the fast kernel (one instantiation per pad count)
            r3 = ops.stack_sigclip(dev(cube, ops), sigma=sig, maxiters=mi, calib=calib, outputs=('mean', 'median', 'std', 'count'))
            assert np.array_equal(host(r3['count']), ref['count']), what + ' planes'
            assert_ulp(host(r3['mean']), ref['mean'].astype(np.float32), 1, what + ' planes: mean')
            assert_ulp(host(r3['median']), ref['median'].astype(np.float32), 1, what + ' planes: median')
            assert_ulp(host(r3['std']), ref['std'].astype(np.float32), 2, what + ' planes: std')
    med = ops.stack_median(dev(cube, ops), calib=calib)
    assert_ulp(host(med), apref.stack_median(cal).astype(np.float32), 0 if N % 2 else 1, f'median N={N}')
    u16 = synth_cube(rng, N, shape, dtype=np.uint16)
    calu = apref.calibrate(u16, bias, dark, nflat, e)
    with np.errstate(all='ignore'):
        ref = apref.stack_sigclip(calu, sigma=3.0, maxiters=5)
    r = ops.stack_sigclip(dev(u16, ops), sigma=3.0, maxiters=5, calib=calib, outputs=('mean', 'count'))
    assert np.array_equal(host(r['count']), ref['count']), f'u16 N={N}'
    assert_ulp(host(r['mean']), ref['mean'].astype(np.float32), 1, f'u16 N={N}')
    r3 = ops.stack_sigclip(dev(u16, ops), sigma=3.0, maxiters=5, calib=calib, outputs=('mean', 'median', 'std', 'count'))
    assert np.array_equal(host(r3['count']), ref['count']), f'u16 planes N={N}'
    assert_ulp(host(r3['mean']), ref['mean'].astype(np.float32), 1, f'u16 planes: mean N={N}')
    assert_ulp(host(r3['median']), ref['median'].astype(np.float32), 1, f'u16 planes: median N={N}')
    assert_ulp(host(r3['std']), ref['std'].astype(np.float32), 2, f'u16 planes: std N={N}')
    assert_ulp(host(ops.stack_median(dev(u16, ops), calib=calib)), apref.stack_median(calu).astype(np.float32),
               0 if N % 2 else 1, f'u16 median N={N}')


def test_sigclip_global_median_neighbours(ops, apref):
    """Even counts need the two middle order statistics; the radix select finds the upper one and takes the lower one from
    (a) the same key, (b) a lower bin of the last digit, (c) a different prefix altogether.  Crafted float32 / float64 columns
    for each case, at sizes with and without full 8192-element pieces, no clipping (sigma 1e300) and with clipping."""
    rng = np.random.default_rng(2024)

    def check(x, what):
        for sigma, maxiters in ((1e300, 1), (3.0, 5)):
            ref = apref.sigclip_global(x, sigma=sigma, maxiters=maxiters)
            s = host(ops.sigclip_global(torch.from_numpy(x).cuda(), sigma=sigma, maxiters=maxiters))
            got = [s[0], s[1], s[2]] if x.dtype == np.float64 else [np.float32(v) for v in s[:3]]
            want = [ref['mean'], ref['median'], ref['std']] if x.dtype == np.float64 else [np.float32(ref[k]) for k in ('mean', 'median', 'std')]
            assert got == want, (what, sigma, got, want)
            assert int(s[6]) == ref['nkeep'] and int(s[5]) == ref['niter'], (what, sigma)

    for dt, it, low_bits in ((np.float32, np.uint32, 10), (np.float64, np.uint64, 9)):
        base = np.array([1.5], dt).view(it)[0]
        aligned = it(int(base) >> low_bits << low_bits)            # smallest key of its last-digit group
        for n in (2, 6, 1000, 8192, 8192 * 3, 8192 * 3 + 10, 40001 * 2):
            half = n // 2
            # (a) the middle pair is one repeated value
            x = np.concatenate([rng.uniform(0.5, 1.0, half - 1), np.full(2, 1.25), rng.uniform(1.5, 2.0, n - half - 1)]).astype(dt)
            check(rng.permutation(x), (dt.__name__, n, 'same key'))
            # (b) lower neighbour a few ulps down, inside the same last-digit group
            hi = np.array([int(aligned) + 7], it).view(dt)[0]
            lo = np.array([int(aligned) + 2], it).view(dt)[0]
            x = np.concatenate([rng.uniform(0.5, 1.0, half - 1), [lo, hi], rng.uniform(1.6, 2.0, n - half - 1)]).astype(dt)
            check(rng.permutation(x), (dt.__name__, n, 'lower bin'))
            # (c) the upper middle value is the first key of its group: the lower one lives under another prefix
            hi = np.array([int(aligned)], it).view(dt)[0]
            x = np.concatenate([rng.uniform(-1.0, 1.0, half), [hi], rng.uniform(1.6, 2.0, n - half - 1)]).astype(dt)
            check(rng.permutation(x), (dt.__name__, n, 'other prefix'))
            # negative / mixed-sign medians and constant data
            check(rng.permutation(np.concatenate([rng.uniform(-3, -1, half), rng.uniform(-1, 4, n - half)]).astype(dt)), (dt.__name__, n, 'mixed'))
            check(np.full(n, -2.75, dt), (dt.__name__, n, 'constant'))


def test_fast32_path_guards_on_adversarial_columns(ops, apref):
    """The float32 fast path of the lean stack kernels (clip_fast32 - full and padded slot counts - / stack_chunks_kernel)
    against the float64 path (exact=True) and the oracle on columns built to hit its guards: means near zero against the spread (the mean-accuracy
    guard), outliers of 1e4..1e30 times the spread, constant and two-level columns, spreads of a few ulp, negative and
    mixed-sign values, values needing more than four trims per side, columns sitting next to a clip bound.  Survivor
    counts must be identical everywhere, means within 1 ulp."""
    rng = np.random.default_rng(2025)
    H, W = 16, 256
    # 13, 30, 37, 52, 58, 75, 100, 119: padded stacks (split pads); 160, 256: chunked kernel; 257 .. 512: chunked kernel, pairs of chunks
    for N in (13, 16, 30, 32, 37, 52, 58, 64, 75, 96, 100, 119, 128, 160, 256, 257, 300, 384, 385, 449, 512):
        cols = []
        base = rng.normal(0.0, 1.0, (N, H, W))
        level = np.array([0.0, 1e-3, 1.0, 50.0, 500.0, 5e4, -300.0, 1e-20])[rng.integers(0, 8, (H, W))]
        spread = np.array([1e-6, 1e-3, 1.0, 30.0, 300.0])[rng.integers(0, 5, (H, W))]
        cube = level[None] + spread[None] * base
        kind = rng.integers(0, 8, (H, W))
        k = min(N // 3, 9 if N <= 256 else 20)
        for f in range(k):                                                          # up to nine (twenty) outliers on one side
            m = (kind == 1) & (rng.random((H, W)) < 0.7)
            cube[f][m] += (10.0 ** rng.uniform(1, 30, m.sum())) * spread[m]
        m = kind == 2
        cube[:, m] = level[m]                                                       # constant columns
        m = kind == 3
        two = rng.random((N, H, W)) < 0.5
        cube[:, m] = np.where(two[:, m], level[m], level[m] + spread[m])            # two discrete levels
        m = kind == 4
        cube[:, m] = np.float32(level[m]) * (1 + np.float32(2.0 ** -22) * rng.integers(-3, 4, (N, m.sum())))   # a few ulp of spread
        m = kind == 5
        cube[:5, m] -= 4.0 * spread[m] * rng.uniform(0.9, 1.6, (5, m.sum()))         # several values near the lower 3-sigma bound
        cube[5:9, m] += 4.0 * spread[m] * rng.uniform(0.9, 1.6, (4, m.sum()))
        cube = cube.astype(np.float32)
        d = dev(cube, ops)
        for sigma, maxiters in ((3.0, 5), (2.0, None), (4.5, 2)):
            with np.errstate(all='ignore'):
                ref = apref.stack_sigclip(cube, sigma=sigma, maxiters=maxiters)
            fast = ops.stack_sigclip(d, sigma=sigma, maxiters=maxiters, outputs=('mean', 'count'))
            exact = ops.stack_sigclip(d, sigma=sigma, maxiters=maxiters, outputs=('mean', 'count'), exact=True)
            what = f'N={N} sigma={sigma} maxiters={maxiters}'
            assert np.array_equal(host(fast['count']), ref['count']), what
            assert np.array_equal(host(exact['count']), ref['count']), what
            assert_ulp(host(exact['mean']), ref['mean'].astype(np.float32), 1, what + ' float64 path')
            assert_ulp(host(fast['mean']), ref['mean'].astype(np.float32), 1, what + ' float32 fast path')
            if N <= 96:
                # the three planes of sigma_clipped_stats on the fast kernel (PLUS): one float64 pass about the fast path's mean
                plus = ops.stack_sigclip(d, sigma=sigma, maxiters=maxiters, outputs=('mean', 'median', 'std', 'count'))
                assert np.array_equal(host(plus['count']), ref['count']), what
                assert_ulp(host(plus['mean']), ref['mean'].astype(np.float32), 1, what + ' planes: mean')
                assert_ulp(host(plus['median']), ref['median'].astype(np.float32), 1, what + ' planes: median')
                assert_ulp(host(plus['std']), ref['std'].astype(np.float32), 2, what + ' planes: std')


def test_unfused_stacks_with_nonfinite_values(ops, apref):
    """Stacks of frames that have already been calibrated or resampled (no fused calibration) hold NaN / +-inf anywhere - a
    resampled frame carries a 6 x 6 block of NaNs around every masked input pixel.  The fast kernel turns them into sentinels,
    starts each lane's clip with its own count of them trimmed (clip_fast32, per-lane form) and hands columns with too many -
    and everything else it cannot finish - to the per-pixel redo list.  Against the oracle and the float64 path: identical
    survivor counts, means within 1 ulp; full and padded slot counts, a pixel mask, a partial last tile, signalling NaNs."""
    rng = np.random.default_rng(4242)
    H, W = 37, 211                                           # 7807 pixels: 30 full tiles + a partial one
    for N in (14, 16, 18, 22, 32, 47, 61, 64, 77, 96, 109, 128):
        cube = synth_cube(rng, N, (H, W))
        # NaN blocks that drift from frame to frame, like the footprints of bad pixels under per-frame shifts
        ys, xs = rng.integers(0, H - 6, 40), rng.integers(0, W - 6, 40)
        for f in range(N):
            dy, dx = rng.integers(-3, 4, 2)
            for y, x in zip(ys, xs):
                yy, xx = min(max(y + dy, 0), H - 6), min(max(x + dx, 0), W - 6)
                cube[f, yy:yy + 6, xx:xx + 6] = np.nan
        bad = rng.random(cube.shape) < 0.004
        cube[bad] = rng.choice(np.array([np.nan, np.inf, -np.inf], np.float32), bad.sum())
        cube[:, 1, 1] = np.nan                                # nothing left
        cube[1:, 1, 2] = np.nan                               # one value left
        cube[: N // 2, 1, 3] = np.inf                         # half the column gone
        for k in range(8):
            cube[:k, 2, k] = -np.inf                          # 0 .. 7 non-finite values in neighbouring columns
        snan = np.array([0x7fa00000], np.uint32).view(np.float32)[0]
        cube.view(np.uint32)[2, 3, 5] = 0x7fa00000            # a signalling NaN
        assert np.isnan(snan)
        pm = (rng.random((H, W)) < 0.03).astype(np.uint8)
        d = dev(cube, ops)
        for sigma, maxiters, mask in ((3.0, 5, None), (2.0, None, pm), (4.0, 1, None)):
            with np.errstate(all='ignore'):
                ref = apref.stack_sigclip(cube, sigma=sigma, maxiters=maxiters, pixmask=mask)
            kw = dict(sigma=sigma, maxiters=maxiters, outputs=('mean', 'count'))
            if mask is not None:
                kw['pixmask'] = dev(mask, ops)
            fast = ops.stack_sigclip(d, **kw)
            exact = ops.stack_sigclip(d, exact=True, **kw)
            what = f'N={N} sigma={sigma} maxiters={maxiters} mask={mask is not None}'
            assert np.array_equal(host(fast['count']), ref['count']), what
            assert np.array_equal(host(exact['count']), ref['count']), what
            assert_ulp(host(exact['mean']), ref['mean'].astype(np.float32), 1, what + ' float64 path')
            assert_ulp(host(fast['mean']), ref['mean'].astype(np.float32), 1, what + ' fast kernel')
        mom = ops.stack_sigclip(d, sigma=3.0, maxiters=5, outputs=('moments',))['moments']
        with np.errstate(all='ignore'):
            ref = apref.stack_sigclip(cube, sigma=3.0, maxiters=5)
        kept = np.where(ref['keep'], np.nan_to_num(cube.astype(np.float64), nan=0.0, posinf=0.0, neginf=0.0), 0.0)
        assert np.array_equal(host(mom)[1], ref['count'].astype(np.float32)), f'moments count N={N}'
        np.testing.assert_allclose(host(mom)[0], kept.sum(0), rtol=3e-7, atol=1e-30)
