"""End-to-end GPU tests of the Ap* shells and ap_* scripts against golden vectors from the reference."""
import json
import os

import numpy as np
import pytest

from tests.util import GOLDEN, assert_biteq, assert_ulp, load_golden, meta, synth_cube

pytestmark = pytest.mark.gpu
torch = pytest.importorskip('torch')


@pytest.fixture(scope='module', autouse=True)
def _need_gpu():
    if not torch.cuda.is_available():
        pytest.skip('no GPU')


def _wf(path, data, **kw):
    from astrophotography_amd import fitsio
    h = fitsio.Header()
    for k, v in kw.items():
        h[k.replace('_', '-')] = v
    fitsio.write(str(path), data, h)


def test_apcalibrate_files_golden(tmp_path):
    """Same FITS files astropy wrote for the reference -> same calibrated pixels and header keywords."""
    import astrophotography_amd as ap
    from astrophotography_amd import fitsio
    g = load_golden('g1_calibrate.npz')
    import shutil
    for n in ('raw', 'bias', 'dark', 'flat', 'cal'):                    # the names the reference run used
        shutil.copy(os.path.join(GOLDEN, f'g1_c0_{n}.fits'), tmp_path / f'{n}.fits')
    src = lambda n: str(tmp_path / f'{n}.fits')
    cal = ap.ApCalibrate(src('bias'), src('dark'), src('flat'), None, 'CRITICAL', dark_still_biased=False)
    out = tmp_path / 'cal_out.fits'
    nf = tmp_path / 'nflat.fits'
    cal.calibrate(src('raw'), str(out), 2, str(nf), False)
    data, hdr = fitsio.read(str(out))
    ref, rhdr = fitsio.read(src('cal'))
    assert data.dtype == np.float32 and hdr['BITPIX'] == -32
    assert_biteq(data, ref, 'calibrated image')
    for k in ('BIASCORR', 'BIASFILE', 'DARKCORR', 'DARKFILE', 'BUNIT', 'FLATCORR', 'FLATFILE', 'EXPTIME'):
        assert hdr[k] == rhdr[k] and hdr.comment(k) == rhdr.comment(k), k
    assert 'BZERO' not in hdr and 'BSCALE' not in hdr
    assert hdr.history()[-1].startswith('Processed by ApCalibrate ')
    assert_biteq(fitsio.read(str(nf))[0], g['nflat_64x64'], 'normalised flat file')
    cal.calibrate(src('raw'), str(out), 2, None, True)               # --fixcosmic: L.A.Cosmic after the repair (test_gpu_lacosmic.py)
    assert fitsio.read(str(out))[1]['CR_CLEAN'] is True
    with pytest.raises(RuntimeError):
        ap.ApCalibrate(src('bias'), str(tmp_path / 'missing.fits'), None, None, 'CRITICAL')


def test_apcalibrate_all_golden_cases(tmp_path):
    import astrophotography_amd as ap
    from astrophotography_amd import fitsio
    g = load_golden('g1_calibrate.npz')
    for ci in range(int(g['ncases'])):
        m = meta(g, f'c{ci}_meta')
        H, W = m['shape']
        shp = f'_{H}x{W}'
        d = tmp_path / f'c{ci}'
        d.mkdir()
        _wf(d / 'bias.fits', g['bias' + shp])
        _wf(d / 'dark.fits', g['dark' + shp], EXPTIME=m['dark_exp'])
        flat = None
        if m['flatmode'] != 'noflat':
            flat = str(d / 'flat.fits')
            _wf(flat, g[m['flatmode'] + shp])
        kw = {m['expkw']: m['img_exp']}
        if m['expkw'] == 'EXPOSURE':
            kw['EXPTIME'] = 999.0
        if m['pedestal'] is not None:
            kw['PEDESTAL'] = m['pedestal']
        _wf(d / 'raw.fits', g['raw_' + m['raw'] + shp], **kw)
        bp = None
        if m['use_mask']:
            bp = str(d / 'bpix.fits')
            _wf(bp, g['mask' + shp])
        cal = ap.ApCalibrate(str(d / 'bias.fits'), str(d / 'dark.fits'), flat, bp, 'CRITICAL', dark_still_biased=m['dark_still_biased'])
        cal.calibrate(str(d / 'raw.fits'), str(d / 'cal.fits'), m['deltapix'], None, False)
        out, hdr = fitsio.read(str(d / 'cal.fits'))
        assert_biteq(out, g[f'c{ci}_out'].astype(np.float32), f'case {ci} {m}')
        ref = {k: v for k, v, _ in json.loads(str(g[f'c{ci}_hdr']))}
        for k in ref:
            if k.startswith('BPIX') or k in ('FLATCORR', 'DARKCORR', 'BIASCORR'):
                assert repr(hdr[k]) == ref[k], (ci, k, hdr[k], ref[k])
        assert 'PEDESTAL' not in hdr
        if ci == 0:      # batch / slab form gives the same file
            cal.calibrate_files([str(d / 'raw.fits')] * 3, [str(d / f'b{i}.fits') for i in range(3)], m['deltapix'])
            assert_biteq(fitsio.read(str(d / 'b2.fits'))[0], out)


def test_apfindbadpixels_golden(tmp_path):
    import astrophotography_amd as ap
    from astrophotography_amd import fitsio
    g = load_golden('g2_findbadpix.npz')
    for ci in range(int(g['ncases'])):
        dark = g[f'd{ci}_dark']
        p = tmp_path / f'dark{ci}.fits'
        _wf(p, dark, TELESCOP='synth', INSTRUME='cam', XBINNING=1)
        fb = ap.ApFindBadPixels(str(p), 4.0, 'CRITICAL')
        assert fb.get_mask().dtype == np.uint8
        if dark.dtype == np.float32:
            assert np.array_equal(fb.get_mask(), g[f'd{ci}_mask_auto'])
            assert fb._nbad_auto == int(g[f'd{ci}_nbad_auto'])
            st = fb.get_stats()
            assert [st['lothresh'], st['hithresh']] == list(g[f'd{ci}_thresh'])
        else:
            # integer dark: numpy's float64 statistics are reproduced exactly on the device
            assert np.array_equal(fb.get_mask(), g[f'd{ci}_mask_auto'])
            st = fb.get_stats()
            assert [float(st['mean']), float(st['median']), float(st['std'])] == list(g[f'd{ci}_stats'])
            assert [st['lothresh'], st['hithresh']] == list(g[f'd{ci}_thresh'])
        if f'd{ci}_mask_user' in g:
            fb.add_user_badpix(os.path.join(GOLDEN, 'user_badpixels.yml'))
            assert np.array_equal(fb.get_mask(), g[f'd{ci}_mask_user'])
            assert fb._nbad_user == int(g[f'd{ci}_nbad_user'])
            mp = tmp_path / 'mask.fits'
            fb.write_mask(str(mp))
            m, hdr = fitsio.read(str(mp))
            assert m.dtype == np.uint8 and np.array_equal(m, g[f'd{ci}_mask_user'])
            ref = {k: v for k, v, _ in json.loads(str(g[f'd{ci}_maskfile_hdr']))}
            for k in ('IMAGETYP', 'CREATOR', 'NBADAUTO', 'NBADUSER', 'USERFILE', 'TELESCOP', 'INSTRUME', 'XBINNING'):
                assert repr(hdr[k]) == ref[k], k


def test_apfixbadpixels_and_imarith_golden(tmp_path):
    import astrophotography_amd as ap
    from astrophotography_amd import fitsio
    g = load_golden('g3_fixbadpix.npz')
    fx = ap.ApFixBadPixels('CRITICAL')
    for dp in (1, 2, 3):
        nd, st = fx.fix_bad_pixels(g['data'], g['mask'], dp)
        assert_biteq(nd, g[f'out_dp{dp}'])
        ref = json.loads(str(g[f'stats_dp{dp}']))
        for k in ref:
            assert st[k][0] == ref[k][0] and st[k][1] == ref[k][1], k
    nd, st = fx.fix_bad_pixels(g['data_u16'], g['mask'], 1)
    assert nd.dtype == np.uint16 and np.array_equal(nd, g['out_u16_dp1'])
    _wf(tmp_path / 'd.fits', g['data'], PEDESTAL=0)
    _wf(tmp_path / 'm.fits', g['mask'])
    fx.fix_files(str(tmp_path / 'd.fits'), str(tmp_path / 'm.fits'), str(tmp_path / 'o.fits'), 2)
    o, h = fitsio.read(str(tmp_path / 'o.fits'))
    assert_biteq(o, g['out_dp2'])
    assert h['BPIXFILE'] == 'm.fits' and h['BPIXDPIX'] == 2 and 'PEDESTAL' not in h

    g = load_golden('g4_imarith.npz')
    ia = ap.ApImArith('CRITICAL')
    _wf(tmp_path / 'a.fits', g['a'], BUNIT='adu')
    _wf(tmp_path / 'b.fits', g['b'])
    _wf(tmp_path / 'au.fits', g['au'])
    _wf(tmp_path / 'bu.fits', g['bu'])
    for op in ('ADD', 'SUB', 'MUL', 'DIV'):
        ia.process_files(str(tmp_path / 'a.fits'), op, str(tmp_path / 'b.fits'), str(tmp_path / 'o.fits'), None)
        assert_biteq(fitsio.read(str(tmp_path / 'o.fits'))[0], g[f'f32_arr_{op}'].astype(np.float32), op)
        ia.process_files(str(tmp_path / 'a.fits'), ' ' + op.lower() + ' ', '3.25', str(tmp_path / 'o.fits'), 'electrons')
        o, h = fitsio.read(str(tmp_path / 'o.fits'))
        assert_biteq(o, g[f'f32_scl_{op}'].astype(np.float32), op)
        assert h['BUNIT'] == 'electrons' and h.history()[-1].endswith(f'a.fits {op} 3.25')
    for op in ('ADD', 'SUB', 'MUL'):
        ia.process_files(str(tmp_path / 'au.fits'), op, str(tmp_path / 'bu.fits'), str(tmp_path / 'o.fits'), None)
        o, _ = fitsio.read(str(tmp_path / 'o.fits'))
        assert o.dtype == np.uint16 and np.array_equal(o, g[f'u16_arr_{op}'])
    # error behaviour recorded from the reference
    for tag, args in (('u16_arr_DIV', ('au.fits', 'DIV', str(tmp_path / 'bu.fits'))), ('u16_scl_ADD', ('au.fits', 'ADD', '2.0')),
                      ('f32_badop', ('a.fits', 'POW', '2.0')), ('f32_badfile', ('a.fits', 'ADD', str(tmp_path / 'none.fits')))):
        exc = {'ValueError': ValueError, 'UFuncTypeError': TypeError, 'TypeError': TypeError}[str(g[tag + '_exc'])]
        with pytest.raises(exc):
            ia.process_files(str(tmp_path / args[0]), args[1], args[2], str(tmp_path / 'o.fits'), None)


def test_apmastercal_and_apstack(tmp_path):
    import astrophotography_amd as ap
    from astrophotography_amd import fitsio
    from oracle import apref
    rng = np.random.default_rng(3)
    N, shape = 12, (40, 64)
    cube = synth_cube(rng, N, shape, dtype=np.uint16)
    d = tmp_path / 'darks'
    d.mkdir()
    for i in range(N):
        _wf(d / f'dark{i:02d}.fits', cube[i], TELESCOP='T05', IMAGETYP='Dark Frame', EXPTIME=300.0, SET_TEMP=-20.0,
            CCD_TEMP=-20.0 + 0.01 * i, DATE_OBS='2020-01-01', FILTER='none')
    mc = ap.ApMasterCal(str(d), 'master*', 'UNKNOWN', 0.5, 'CRITICAL')
    mc.make_master(str(d / 'master_dark.fits'))
    m, h = fitsio.read(str(d / 'master_dark.fits'))
    ref = apref.combine_ccdproc(cube.astype(np.float64), 5.0, 5.0)
    assert m.dtype == np.float64 and h['BITPIX'] == -64           # CCDData product: float64 primary + MASK + UNCERT
    # the float64 mean is written as it is (not narrowed to float32 on the device): float64 agreement with the oracle,
    # and the values are NOT all float32-representable
    np.testing.assert_allclose(m, ref['mean'], rtol=1e-14)
    assert (m != m.astype(np.float32)).any()
    mask, mh = fitsio.read_extension(str(d / 'master_dark.fits'), 'MASK')
    unc, uh = fitsio.read_extension(str(d / 'master_dark.fits'), 'UNCERT')
    assert mask.dtype == np.uint8 and mask.shape == shape and not mask.any()
    assert uh['UTYPE'] == 'StdDevUncertainty' and unc.dtype == np.float64
    np.testing.assert_allclose(unc, ref['std'] / np.sqrt(ref['count']), rtol=1e-12, atol=1e-300)
    assert h['IMAGETYP'] == 'MASTER DARK' and h['NCOMBINE'] == N and h['IFILE011'] == 'dark11.fits' and h['BUNIT'] == 'adu'
    # second construction ignores the master it just wrote
    assert len(ap.ApMasterCal(str(d), 'master*', 'UNKNOWN', 0.5, 'CRITICAL')._values('file')) == N
    # the script front-end
    from astrophotography_amd.scripts import ap_combine_darks, ap_stack
    assert ap_combine_darks.main([str(d), str(d / 'master2.fits'), '-l', 'CRITICAL']) == 0
    assert np.array_equal(fitsio.read(str(d / 'master2.fits'))[0], m)
    assert ap_combine_darks.main([str(tmp_path / 'nodir'), str(d / 'x.fits'), '-l', 'CRITICAL']) == 1

    # ApStack on files, fused with calibration, vs the oracle
    bias = rng.normal(100, 2, shape).astype(np.float32)
    dark = rng.normal(10, 1, shape).astype(np.float32)
    flat = rng.normal(30000, 300, shape).astype(np.float32)
    _wf(tmp_path / 'bias.fits', bias)
    _wf(tmp_path / 'dark.fits', dark, EXPTIME=300.0)
    _wf(tmp_path / 'flat.fits', flat)
    files = []
    for i in range(N):
        p = tmp_path / f'raw{i:02d}.fits'
        _wf(p, cube[i], EXPTIME=120.0)
        files.append(str(p))
    assert ap_stack.main([str(tmp_path / 'stack.fits')] + files + ['--master_bias', str(tmp_path / 'bias.fits'), '--master_dark',
                         str(tmp_path / 'dark.fits'), '--master_flat', str(tmp_path / 'flat.fits'), '-l', 'CRITICAL']) == 0
    s, hs = fitsio.read(str(tmp_path / 'stack.fits'))
    nflat, _ = apref.flat_normalize(flat)
    ref_mean, _ = apref.calibrate_stack(cube, bias, dark, nflat, 120.0 / 300.0, sigma=3.0, maxiters=5)
    assert_ulp(s, ref_mean, 1, 'ap_stack fused')
    assert hs['NCOMBINE'] == N and hs['STACKMET'] == 'sigclip'
    st = ap.ApStack('CRITICAL')
    r = st.stack(torch.from_numpy(cube.astype(np.float32)).cuda(), method='median')
    assert_ulp(r['median'].cpu().numpy(), apref.stack_median(cube.astype(np.float32)).astype(np.float32), 1, 'median')
    r = st.stack(torch.from_numpy(cube.astype(np.float32)).cuda(), method='mean')
    np.testing.assert_allclose(r['mean'].cpu().numpy(), cube.astype(np.float64).mean(0), rtol=2e-7)
    with pytest.raises(ValueError):
        st.stack(torch.zeros((2, 4, 4), device='cuda'), method='mode')


def test_stack_mad_std_vs_oracle():
    from astrophotography_amd import ops
    from oracle import apref
    rng = np.random.default_rng(29)
    for N in (3, 8, 16, 33, 64):
        cube = synth_cube(rng, N, (23, 31), nan_frac=0.01)
        cube[:, 0, 0] = 5.0
        d = torch.from_numpy(cube).cuda()
        for sigma, maxiters, cen in ((5.0, 1, 'median'), (3.0, 5, 'median'), (3.0, None, 'mean')):
            ref = apref.stack_sigclip(cube, sigma=sigma, maxiters=maxiters, cenfunc=cen, stdfunc='mad_std')
            r = ops.stack_sigclip(d, sigma=sigma, maxiters=maxiters, cenfunc=cen, stdfunc='mad_std', outputs=('mean', 'count'))
            what = f'N={N} sigma={sigma} maxiters={maxiters} cen={cen}'
            assert np.array_equal(r['count'].cpu().numpy(), ref['count']), what
            assert_ulp(r['mean'].cpu().numpy(), ref['mean'].astype(np.float32), 1, what)
        for form, flag in (('legacy', False), ('astropy', True)):
            c = apref.combine_ccdproc(cube, 5.0, 5.0, form=form)
            r = ops.stack_sigclip(d, sigma=5.0, maxiters=1, cenfunc='median', stdfunc='mad_std', outputs=('mean', 'count'),
                                  nonfinite_unclipped=flag)
            assert np.array_equal(r['count'].cpu().numpy(), c['count']), (N, form)


def test_nshard_collective_path_on_one_gpu():
    """The striped all-reduce path of parallel.stack_nshard, exercised with an RCCL group of one rank:
    the result must equal the direct single-kernel result (world_size 1 => identical semantics)."""
    import socket
    import torch.distributed as dist
    from astrophotography_amd import ops, parallel, synth
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    dist.init_process_group('nccl', init_method=f'tcp://127.0.0.1:{port}', rank=0, world_size=1,
                            device_id=torch.device('cuda', 0))
    try:
        N, H, W = 16, 96, 256
        masters = synth.make_masters(H, W, config_id=3, device='cuda')
        nflat, _ = ops.flat_normalize(masters['flat'])
        frames = synth.make_frames(N, masters, nflat, config_id=3)
        calib = dict(bias=masters['bias'], dark=masters['dark'], nflat=nflat, exp_ratio=synth.EXP_RATIO)
        direct = ops.stack_sigclip(frames, calib=calib, outputs=('mean', 'std', 'count', 'moments'))
        direct64 = ops.stack_sigclip(frames, calib=calib, outputs=('moments_f64p',), moments_mean_only=True)['moments_f64p']
        legacy = ops.stack_sigclip(frames, calib=calib, outputs=('moments_f64',))['moments_f64']     # float64 path (no flag)
        assert torch.equal(legacy['count'], direct['count']) and torch.equal(direct64['count'].to(torch.int32), direct['count'])
        assert float(((legacy['sum'] - direct64['sum']).abs() / legacy['sum'].abs()).max()) < 1e-7   # float32 vs float64 sums
        for exchange in ('f64', 'f32', 'rs'):
            # ('rs' with 5 stripes of 96 rows: ragged stripes exist, and with ONE rank every stripe divides - the
            #  reduce-scatter / finalise-own-rows / all-gather form runs with real kernels and an RCCL group)
            mean, parts = parallel.stack_nshard(frames, calib, n_stripes=5, force_collective=True, return_moments=True,
                                                exchange=exchange)
            torch.cuda.synchronize()
            if exchange == 'f32':
                assert torch.equal(torch.cat([p['sum'] for p in parts], 0), direct['moments'][0])
                assert torch.equal(torch.cat([p['count'] for p in parts], 0), direct['moments'][1])
            else:
                assert torch.equal(torch.cat([p['sum'] for p in parts], 0), direct64['sum'])
                assert torch.equal(torch.cat([p['count'] for p in parts], 0).to(torch.int32), direct['count'])
            # mean from the moments (sum / count) vs the kernel's float64 mean c + S/n: within 1 ulp
            assert_ulp(mean.cpu().numpy(), direct['mean'].cpu().numpy(), 1, 'moments-finalised mean ' + exchange)
        for exchange in ('rs', 'f64'):
            mean, std = parallel.stack_nshard(frames, calib, n_stripes=3, force_collective=True, want_std=True, exchange=exchange)
            torch.cuda.synchronize()
            assert_ulp(mean.cpu().numpy(), direct['mean'].cpu().numpy(), 1, 'mean with std ' + exchange)
            assert float(((std - direct['std']).abs() / direct['std']).max()) < 1e-5
        # exact=True: the float64 clip only - the moments are the float64 sums of the survivors (legacy layout's values)
        mean_x, parts_x = parallel.stack_nshard(frames, calib, n_stripes=2, force_collective=True, return_moments=True, exchange='f64', exact=True)
        torch.cuda.synchronize()
        assert torch.equal(torch.cat([p['sum'] for p in parts_x], 0), legacy['sum'])
        # row-sharded exact path: two half images reduce to the same pixels
        top = parallel.stack_rowshard(frames[:, :48], dict(calib, bias=calib['bias'][:48], dark=calib['dark'][:48], nflat=nflat[:48]))
        assert torch.equal(top['mean'], direct['mean'][:48])
    finally:
        dist.destroy_process_group()


def test_fits_device_decode_encode(tmp_path):
    """F1: big-endian payload decode/encode on the device equals the host (numpy) reader/writer bit for bit."""
    from astrophotography_amd import fitsio
    rng = np.random.default_rng(5)
    cases = {
        'u16': rng.integers(0, 65536, (37, 53)).astype(np.uint16),         # odd pixel count
        'u16b': rng.integers(0, 65536, (64, 64)).astype(np.uint16),
        'i16': rng.integers(-32768, 32767, (20, 31)).astype(np.int16),
        'f32': rng.normal(0, 1e3, (33, 47)).astype(np.float32),
        'f64': rng.normal(0, 1, (8, 9)),
        'u8': rng.integers(0, 4, (10, 12)).astype(np.uint8),
    }
    cases['f32'][0, 0] = np.nan
    cases['f32'][0, 1] = -0.0
    for name, arr in cases.items():
        p = tmp_path / f'{name}.fits'
        h = fitsio.Header()
        h['EXPTIME'] = 12.5
        fitsio.write(str(p), arr, h)
        t, hd = fitsio.read_device(str(p))
        ref, hr = fitsio.read(str(p))
        assert hd['EXPTIME'] == 12.5 and hd.keys() == hr.keys()
        if name.startswith('u16'):
            assert t.dtype == torch.uint16
            assert np.array_equal(t.view(torch.int16).cpu().numpy().view(np.uint16), ref)
        elif name == 'i16':
            assert t.dtype == torch.float32 and np.array_equal(t.cpu().numpy(), ref.astype(np.float32))
        else:
            assert_biteq(t.cpu().numpy(), ref, name)
    t = torch.from_numpy(cases['f32']).cuda()
    _, h = fitsio.read(str(tmp_path / 'f32.fits'))
    fitsio.write_device(str(tmp_path / 'dev.fits'), t, h)
    assert (tmp_path / 'dev.fits').read_bytes() == (tmp_path / 'f32.fits').read_bytes()


def test_read_noise_golden(tmp_path, capsys):
    """F2: ApImageDifference / ApCalcReadNoise against the reference's own numbers (exact float64 equality)."""
    import astrophotography_amd as ap
    g = load_golden('g8_readnoise.npz')
    for tag in ('u16', 'f32'):
        b1, b2 = g[tag + '_b1'], g[tag + '_b2']
        for clip in (1, 0):
            d = ap.ApImageDifference(b1, b2, bool(clip), 'CRITICAL')
            ref = g[f'{tag}_clip{clip}_stats']
            ng, nt = d.numpix()
            assert [d.stddev(), d.min(), d.max(), d.mean(), d.median(), ng, nt] == list(ref), (tag, clip)
            good = np.unpackbits(g[f'{tag}_clip{clip}_good'])[:b1.size].reshape(b1.shape).astype(bool)
            assert np.array_equal(d.good_pixel_mask(), good)
        d = ap.ApImageDifference(b1, b2, False, 'CRITICAL', mask1=g[tag + '_mask1'])
        assert [d.stddev(), d.min(), d.max(), d.mean(), d.median(), d.numpix()[0]] == list(g[tag + '_masked_stats'])
        assert np.array_equal(d.data(), b1.astype(np.float64) - b2.astype(np.float64))
        _wf(tmp_path / f'{tag}1.fits', b1, EGAIN=1.37)
        _wf(tmp_path / f'{tag}2.fits', b2, EGAIN=1.37)
        rn = ap.ApCalcReadNoise(str(tmp_path / f'{tag}1.fits'), str(tmp_path / f'{tag}2.fits'), 'EGAIN', 'CRITICAL').estimate_rn(True)
        rn2 = ap.ApCalcReadNoise(str(tmp_path / f'{tag}1.fits'), str(tmp_path / f'{tag}2.fits'), '2.0', 'CRITICAL').estimate_rn(False)
        assert [rn, rn2] == list(g[tag + '_readnoise'])
    from astrophotography_amd.scripts import ap_calc_read_noise
    assert ap_calc_read_noise.main([str(tmp_path / 'u161.fits'), str(tmp_path / 'u162.fits'), '-l', 'CRITICAL']) == 0
    assert 'Estimated read noise is %.2f electrons/pixel.' % g['u16_readnoise'][0] in capsys.readouterr().out
    with pytest.raises(RuntimeError):
        ap.ApCalcReadNoise(str(tmp_path / 'u161.fits'), str(tmp_path / 'u162.fits'), 'NOGAIN', 'CRITICAL').estimate_rn(True)
    with pytest.raises(RuntimeError):
        ap.ApImageDifference(b1, b2[:10], True, 'CRITICAL')


def test_apresample_files_and_script(tmp_path):
    """F3 at file level: ApResample / ap_coadd.py (the SWarp step of resample_all.sh) against 'oracle resample,
    then numpy combine'; FSCALE = 1/EXPOSURE except in SUM mode; weight image = contributing frames."""
    import warnings
    import yaml
    import astrophotography_amd as ap
    from astrophotography_amd import fitsio
    from astrophotography_amd.scripts import ap_coadd
    from oracle import apref
    rng = np.random.default_rng(21)
    N, shape = 6, (48, 72)
    cube = rng.normal(400, 12, (N,) + shape).astype(np.float32)
    exps = [60.0, 60.0, 120.0, 120.0, 30.0, 90.0]
    A = []
    names = []
    for i in range(N):
        th = np.deg2rad(rng.uniform(-0.3, 0.3))
        A.append([np.cos(th), -np.sin(th), rng.uniform(-2, 2), np.sin(th), np.cos(th), rng.uniform(-2, 2)])
        kw = dict(EXPOSURE=exps[i]) if i % 2 == 0 else dict(EXPTIME=exps[i])
        _wf(tmp_path / f'cal{i}.fits', cube[i], **kw)
        names.append(str(tmp_path / f'cal{i}.fits'))
    mask = (rng.random(shape) < 0.004).astype(np.uint8)
    _wf(tmp_path / 'badpix.fits', mask)
    fs = np.array([1.0 / e for e in exps], np.float32)
    res_ref, _ = apref.resample_affine(cube, A, fscale=fs, mask=mask, conserve_flux=True)
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        med = np.nanmedian(res_ref.astype(np.float64), axis=0).astype(np.float32)
    rs = ap.ApResample('CRITICAL', combine='MEDIAN')
    rs.coadd_files(names, A, str(tmp_path / 'coadd.fits'), weight_file=str(tmp_path / 'w.fits'), mask_file=str(tmp_path / 'badpix.fits'))
    img, h = fitsio.read(str(tmp_path / 'coadd.fits'))
    wimg, _ = fitsio.read(str(tmp_path / 'w.fits'))
    assert_ulp(img, med, 1, 'median co-add file')
    assert np.array_equal(wimg, np.isfinite(res_ref).sum(0).astype(np.float32))
    assert h['NCOMBINE'] == N and h['COMBINET'] == 'MEDIAN' and h['TEXPTIME'] == sum(exps) and h['IFILE005'] == 'cal5.fits'
    # SUM mode does not flux-scale (resample_all.sh:305-309)
    res_sum, _ = apref.resample_affine(cube, A, mask=mask, conserve_flux=True)
    with open(tmp_path / 't.yml', 'w') as fh:
        yaml.safe_dump({'transforms': {f'cal{i}.fits': [float(v) for v in A[i]] for i in range(N)}}, fh)
    assert ap_coadd.main([str(tmp_path / 'sum.fits'), *names, '--transforms', str(tmp_path / 't.yml'), '--combine', 'SUM',
                          '--badpix', str(tmp_path / 'badpix.fits'), '-l', 'CRITICAL']) == 0
    s, _ = fitsio.read(str(tmp_path / 'sum.fits'))
    ok = np.isfinite(res_sum).any(0)
    assert_ulp(s[ok], np.nansum(res_sum.astype(np.float64), axis=0)[ok].astype(np.float32), 1, 'sum co-add file')
    with pytest.raises(RuntimeError):
        rs.coadd_files(names, A[:-1], str(tmp_path / 'x.fits'))
    with pytest.raises(ValueError):
        ap.ApResample('CRITICAL', combine='MODE')


def test_read_slab_device_and_float64_masters(tmp_path):
    """F1 for the stackers: fitsio.read_slab_device (pinned double-buffered staging, payload decoded on the device straight
    into the slab) against the host reader for every layout; ApMasterCal on FLOAT64 frames = the oracle's float64 combine
    bit for bit (golden group G12, nothing narrowed to float32); ApStack.stack_files with a calibrator built from the
    float64 masters ApMasterCal writes."""
    import json
    import astrophotography_amd as ap
    from astrophotography_amd import fitsio, ops
    from oracle import apref
    from tests.util import load_golden
    rng = np.random.default_rng(77)
    shape = (37, 53)
    u16 = rng.integers(0, 65535, (5,) + shape).astype(np.uint16)
    f32 = rng.normal(500, 30, (5,) + shape).astype(np.float32)
    f64 = rng.normal(500, 30, (3,) + shape)
    i16 = rng.integers(-3000, 3000, (2,) + shape).astype(np.int16)
    names = {}
    for tag, cube in (('u16', u16), ('f32', f32), ('f64', f64), ('i16', i16)):
        names[tag] = []
        for i in range(cube.shape[0]):
            p = tmp_path / f'{tag}_{i}.fits'
            _wf(p, cube[i], EXPTIME=10.0 + i)
            names[tag].append(str(p))
    tm = {}
    slab, hdrs = fitsio.read_slab_device(names['u16'], timings=tm)
    assert slab.dtype == torch.uint16 and np.array_equal(slab.view(torch.int16).cpu().numpy().view(np.uint16), u16)
    assert [h['EXPTIME'] for h in hdrs] == [10.0 + i for i in range(5)] and tm['total'] >= tm['read'] >= 0
    slab, _ = fitsio.read_slab_device(names['f32'])
    assert slab.dtype == torch.float32 and np.array_equal(slab.cpu().numpy(), f32)
    slab, _ = fitsio.read_slab_device(names['f64'])
    assert slab.dtype == torch.float64 and np.array_equal(slab.cpu().numpy(), f64)
    slab, _ = fitsio.read_slab_device(names['i16'] + names['u16'][:2])                 # mixed integers widen exactly
    assert slab.dtype == torch.float32 and np.array_equal(slab.cpu().numpy(), np.concatenate([i16, u16[:2]]).astype(np.float32))
    slab, _ = fitsio.read_slab_device(names['f32'][:2] + names['f64'][:1])             # any float64 file: nothing is narrowed
    assert slab.dtype == torch.float64 and np.array_equal(slab.cpu().numpy()[2], f64[0])
    slab, _ = fitsio.read_slab_device(names['u16'], dtype=torch.float32)
    assert np.array_equal(slab.cpu().numpy(), u16.astype(np.float32))
    with pytest.raises(RuntimeError):
        _wf(tmp_path / 'odd.fits', np.zeros((5, 5), np.float32))
        fitsio.read_slab_device(names['f32'][:1] + [str(tmp_path / 'odd.fits')])

    # ApMasterCal on float64 frames: the float64 combine kernel, bit for bit the oracle / G12
    g = load_golden('g12_combine.npz')
    case = [m for m in json.loads(str(g['_meta'])) if m['kind'] == 'f64ties' and m['N'] == 16][0]['case']
    frames = g[f'c{case}_frames']
    d = tmp_path / 'd64'
    d.mkdir()
    for i in range(frames.shape[0]):
        _wf(d / f'dark{i:02d}.fits', frames[i], TELESCOP='T05', IMAGETYP='Dark Frame', EXPTIME=300.0, SET_TEMP=-20.0,
            CCD_TEMP=-20.0, DATE_OBS='2020-01-01', FILTER='none')
    ap.ApMasterCal(str(d), 'master*', 'UNKNOWN', 0.5, 'CRITICAL').make_master(str(d / 'master_dark.fits'))
    m, h = fitsio.read(str(d / 'master_dark.fits'))
    assert m.dtype == np.float64
    assert_biteq(m, g[f'c{case}_mean'])
    unc, _ = fitsio.read_extension(str(d / 'master_dark.fits'), 'UNCERT')
    assert_biteq(unc, g[f'c{case}_std'] / np.sqrt(g[f'c{case}_count'].astype(np.float64)))
    r = ops.combine_f64(torch.from_numpy(frames).cuda())
    assert np.array_equal(r['count'].cpu().numpy(), g[f'c{case}_count'])                # the values ON the +-5 dev bounds are kept
    ref = apref.combine_ccdproc(frames, 5.0, 5.0)
    assert_biteq(r['mean_f64'].cpu().numpy(), ref['mean'])
    assert_biteq(r['std_f64'].cpu().numpy(), ref['std'])
    # both published forms on the float64 kernel, every float64 case of G12 - including the columns on which they disagree
    # (values within an ulp of base +- 5 dev: group f64bounds) - bit for bit the golden arrays (numpy.ma + astropy run for real)
    nd = 0
    for mm in json.loads(str(g['_meta'])):
        if mm['kind'] not in ('f64ties', 'f64bounds'):
            continue
        kk = mm['case']
        fr = torch.from_numpy(g[f'c{kk}_frames']).cuda()
        for form, tag in (('legacy', ''), ('astropy', 'b_')):
            rr = ops.combine_f64(fr, form=form)
            assert np.array_equal(rr['count'].cpu().numpy(), g[f'c{kk}_{tag}count']), (mm, form)
            assert_biteq(rr['mean_f64'].cpu().numpy(), g[f'c{kk}_{tag}mean'])
            assert_biteq(rr['std_f64'].cpu().numpy(), g[f'c{kk}_{tag}std'])
        nd += int((g[f'c{kk}_count'] != g[f'c{kk}_b_count']).sum())
    assert nd >= 20

    # float64 masters (what ApMasterCal writes) -> ApCalibrate -> ApStack.stack_files(calibrator=...)
    N = 8
    raw = np.clip(rng.normal(1500, 40, (N,) + shape), 0, 65535).astype(np.uint16)
    bias, dark = rng.normal(100, 2, shape), rng.normal(10, 1, shape)                   # float64
    flat = rng.normal(30000, 300, shape)
    _wf(tmp_path / 'mb.fits', bias)
    _wf(tmp_path / 'md.fits', dark, EXPTIME=300.0)
    _wf(tmp_path / 'mf.fits', flat)
    files = []
    for i in range(N):
        _wf(tmp_path / f'light{i}.fits', raw[i], EXPTIME=120.0)
        files.append(str(tmp_path / f'light{i}.fits'))
    cal = ap.ApCalibrate(str(tmp_path / 'mb.fits'), str(tmp_path / 'md.fits'), str(tmp_path / 'mf.fits'), None, 'CRITICAL',
                         dark_still_biased=False)
    res = ap.ApStack('CRITICAL').stack_files(files, str(tmp_path / 'stk.fits'), calibrator=cal)
    # reference semantics: NumPy calibrates in float64 when a master is float64 (ApCalibrate.py:439-464); the calibrated
    # frames are then stacked (float32 stack kernel: values rounded to float32)
    nflat = flat / np.nanmean(flat)
    calf = ((raw.astype(np.float32) - bias) - (120.0 / 300.0) * dark) / nflat
    ref = apref.stack_sigclip(calf.astype(np.float32), sigma=3.0, maxiters=5)
    s, _ = fitsio.read(str(tmp_path / 'stk.fits'))
    assert np.array_equal(res['count'].cpu().numpy(), ref['count'])
    assert_ulp(s, ref['mean'].astype(np.float32), 1, 'stack_files with float64 masters')
