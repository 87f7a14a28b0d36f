"""Randomised parity sweep (fixed seeds): many small stacks of random shape / frame count / dtype / options through
every kernel family (lean, rich, median; one- and two-pixel-per-lane uint16 paths; full and padded slot counts;
row stripes of a larger slab) against the oracle.  Catches dispatch and alignment corner cases."""
import numpy as np
import pytest

from tests.util import assert_ulp, synth_cube, synth_masters

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


def _dev(a, ops):
    a = np.ascontiguousarray(a)
    return ops.to_device_u16(a) if a.dtype == np.uint16 else torch.from_numpy(a).cuda()


@pytest.mark.parametrize('seed', range(24))
def test_random_stack_configs(ops, apref, seed):
    rng = np.random.default_rng(10_000 + seed)
    N = int(rng.choice([1, 2, 3, 4, 5, 7, 8, 9, 11, 12, 13, 16, 17, 23, 24, 25, 31, 32, 33, 47, 48, 49, 63, 64, 65, 95, 96, 97, 127, 128]))
    if rng.integers(0, 2):
        N = int(rng.integers(1, 129))                       # every pad count of every slot count (the fast kernels are per pad count)
    H, W = int(rng.integers(1, 9)), int(rng.integers(1, 300))
    u16 = bool(rng.integers(0, 2))
    cube = synth_cube(rng, N, (H, W), nan_frac=0.0 if u16 else 0.02, dtype=np.uint16 if u16 else np.float32)
    use_calib = bool(rng.integers(0, 4) > 0)
    bias, dark, flat = synth_masters(rng, (H, W))
    nflat = (flat / np.float32(30000.0)).astype(np.float32) if rng.integers(0, 3) > 0 else None
    if nflat is not None and W > 3:
        nflat[0, :3] = [0.0, -1.5, np.nan]
    e_uniform = bool(rng.integers(0, 3) > 0)
    e = np.full(N, 0.4, np.float32) if e_uniform else rng.uniform(0.2, 0.6, N).astype(np.float32)
    sb = bool(rng.integers(0, 2))
    if use_calib:
        cal = apref.calibrate(cube, bias, dark, nflat, e, dark_still_biased=sb)
        calib = dict(bias=_dev(bias, ops), dark=_dev(dark, ops), nflat=None if nflat is None else _dev(nflat, ops),
                     exp_ratio=_dev(e, ops), dark_still_biased=sb)
    else:
        cal = cube.astype(np.float32)
        calib = None
    pixmask = (rng.random((H, W)) < 0.05).astype(np.uint8) if rng.integers(0, 2) else None
    pm = None if pixmask is None else _dev(pixmask, ops)
    sigma = float(rng.choice([2.0, 3.0, 5.0]))
    maxiters = rng.choice([1, 5, None])
    maxiters = None if maxiters is None else int(maxiters)
    cen = str(rng.choice(['median', 'mean']))
    dv = str(rng.choice(['std', 'std', 'mad_std']))
    what = f'seed={seed} N={N} {H}x{W} u16={u16} calib={use_calib} flat={nflat is not None} e_uniform={e_uniform} {cen}/{dv} s={sigma} it={maxiters}'
    with np.errstate(all='ignore'):
        ref = apref.stack_sigclip(cal, sigma=sigma, maxiters=maxiters, cenfunc=cen, stdfunc=dv)
    mref, nref = ref['mean'].astype(np.float32), ref['count'].copy()
    if pixmask is not None:
        mref[pixmask != 0] = np.nan
        nref[pixmask != 0] = 0
    outs = ('mean', 'count') if dv == 'std' and rng.integers(0, 2) else ('mean', 'count', 'median', 'std')
    r = ops.stack_sigclip(_dev(cube, ops), sigma=sigma, maxiters=maxiters, cenfunc=cen, stdfunc=dv, calib=calib, pixmask=pm,
                          outputs=outs)
    # Exact ties: integer-valued frames whose calibration only shifts a column (no flat, one exposure ratio)
    # can put a value exactly ON a clip bound, where astropy's own answer hangs on the last bit of its frame-order sums.  A
    # pixel is on a tie when the oracle itself changes its count under a 1e-9 relative change of sigma; such pixels (found by
    # the 20-minute run of round 3: 1 pixel in 6000 cases) are left out.
    got_cnt, got_mean = r['count'].cpu().numpy(), r['mean'].cpu().numpy()
    tie = np.zeros((H, W), bool)
    for (y, x) in np.argwhere(got_cnt != nref)[:8]:
        with np.errstate(all='ignore'):
            r_lo = apref.stack_sigclip(cal[:, y:y + 1, x:x + 1], sigma=sigma * (1 - 1e-9), maxiters=maxiters, cenfunc=cen, stdfunc=dv)
            r_hi = apref.stack_sigclip(cal[:, y:y + 1, x:x + 1], sigma=sigma * (1 + 1e-9), maxiters=maxiters, cenfunc=cen, stdfunc=dv)
        c_lo, c_hi = r_lo['count'][0, 0], r_hi['count'][0, 0]
        tie[y, x] = c_lo != c_hi
        if tie[y, x]:
            # a tie pixel is not a free pass: the kernel's answer must be the oracle's on ONE side of the tie
            side = r_lo if got_cnt[y, x] == c_lo else r_hi
            assert got_cnt[y, x] == side['count'][0, 0], 'tie pixel count matches neither side ' + what
            assert_ulp(got_mean[y:y + 1, x:x + 1], side['mean'].astype(np.float32), 1, 'tie pixel mean ' + what)
    assert tie.sum() <= 2, 'too many tie pixels ' + what
    if dv == 'std' and cen == 'median' and outs == ('mean', 'count'):
        # the float32 fast path claims the survivor sets of the float64 path by construction: fast == exact, every pixel
        rx = ops.stack_sigclip(_dev(cube, ops), sigma=sigma, maxiters=maxiters, cenfunc=cen, stdfunc=dv, calib=calib, pixmask=pm,
                               outputs=outs, exact=True)
        assert np.array_equal(rx['count'].cpu().numpy(), got_cnt), 'fast32 vs exact counts ' + what
        assert_ulp(rx['mean'].cpu().numpy(), got_mean, 1, 'fast32 vs exact mean ' + what)
    assert np.array_equal(got_cnt[~tie], nref[~tie]), what
    assert_ulp(got_mean[~tie], mref[~tie], 1, what)
    med, cnt = ops.stack_median(_dev(cube, ops), calib=calib, pixmask=pm, want_count=True)
    mm = apref.stack_median(cal).astype(np.float32)
    nn = (~np.isnan(cal)).sum(0).astype(np.int32)
    if pixmask is not None:
        mm[pixmask != 0] = np.nan
        nn[pixmask != 0] = 0
    assert np.array_equal(cnt.cpu().numpy(), nn), what
    assert_ulp(med.cpu().numpy(), mm, 1, 'median ' + what)
    # a row stripe of the slab (frame_stride > stripe pixels), as the N-shard path uses it
    if H >= 3:
        r0, r1 = 1, H - 1
        sub = _dev(cube, ops)[:, r0:r1]
        c2 = None
        if calib is not None:
            c2 = dict(calib)
            for k in ('bias', 'dark', 'nflat'):
                if c2.get(k) is not None:
                    c2[k] = c2[k][r0:r1]
        rs = ops.stack_sigclip(sub, sigma=sigma, maxiters=maxiters, cenfunc=cen, stdfunc=dv, calib=c2,
                               pixmask=None if pm is None else pm[r0:r1], outputs=('mean', 'count'))
        assert np.array_equal(rs['count'].cpu().numpy()[~tie[r0:r1]], nref[r0:r1][~tie[r0:r1]]), 'stripe ' + what
        assert_ulp(rs['mean'].cpu().numpy()[~tie[r0:r1]], mref[r0:r1][~tie[r0:r1]], 1, 'stripe ' + what)


def test_too_many_frames_is_refused(ops):
    from astrophotography_amd._lib import ApGpuError
    with pytest.raises(ApGpuError):
        ops.stack_sigclip(torch.zeros((513, 2, 8), device='cuda'))


@pytest.mark.parametrize('seed', range(12))
def test_random_image_kernels(ops, apref, seed):
    """Streaming / statistics kernels on random (often awkward) shapes: vector tails, sizes below one reduction
    piece, ragged last pieces, masks that are empty or full."""
    rng = np.random.default_rng(20_000 + seed)
    H, W = int(rng.integers(1, 70)), int(rng.integers(1, 400))
    N = int(rng.integers(1, 6))
    img = rng.normal(50, 6, (H, W)).astype(np.float32)
    img[rng.random((H, W)) < 0.01] += 500
    bias, dark, flat = synth_masters(rng, (H, W))
    nflat_ref, norm_ref = apref.flat_normalize(flat)
    nflat, norm = ops.flat_normalize(_dev(flat, ops))
    assert np.float32(norm.item()) == norm_ref and np.array_equal(nflat.cpu().numpy(), nflat_ref), (seed, H, W)
    u16 = bool(rng.integers(0, 2))
    raw = synth_cube(rng, N, (H, W), dtype=np.uint16 if u16 else np.float32)
    e = rng.uniform(0.2, 0.6, N).astype(np.float32)
    ped = rng.choice([0.0, -100.0], N).astype(np.float32) if rng.integers(0, 2) else None
    ref = apref.calibrate(raw, bias, dark, nflat_ref, e, pedestal=ped)
    got = ops.calibrate(_dev(raw, ops), _dev(bias, ops), _dev(dark, ops), nflat, _dev(e, ops), None if ped is None else _dev(ped, ops))
    assert np.array_equal(got.cpu().numpy(), ref, equal_nan=True), ('calibrate', seed, H, W, N, u16)
    # global clip + threshold mask + repair
    st_ref = apref.sigclip_global(img, sigma=4.0, maxiters=5)
    st = ops.sigclip_global(_dev(img, ops), sigma=4.0, maxiters=5).cpu().numpy()
    for k, name in enumerate(('mean', 'median', 'std')):
        assert np.float32(st[k]) == np.float32(st_ref[name]), (name, seed, H, W, st[k], st_ref[name])
    lo, hi = apref.badpix_thresholds(st_ref['median'], st_ref['std'], 4.0)
    mref = apref.threshold_mask(img, lo, hi)
    mask, nbad = ops.threshold_mask(_dev(img, ops), lo, hi)
    mref_arr = mref[0] if isinstance(mref, tuple) else mref
    assert np.array_equal(mask.cpu().numpy(), mref_arr) and int(nbad.item()) == int(mref_arr.sum())
    delta = int(rng.integers(1, 4))
    fref = apref.fix_badpix(img, mref_arr, delta)
    fgot = ops.fix_badpix(_dev(img, ops), mask, delta)
    fref_arr = fref[0] if isinstance(fref, tuple) else fref
    fgot_arr = fgot[0] if isinstance(fgot, tuple) else fgot
    assert np.array_equal(fgot_arr.cpu().numpy(), fref_arr, equal_nan=True), ('fix_badpix', seed, H, W, delta)
    # image arithmetic
    for op in ('ADD', 'SUB', 'MUL', 'DIV'):
        with np.errstate(all='ignore'):
            assert np.array_equal(ops.imarith(_dev(img, ops), op, _dev(bias, ops)).cpu().numpy(), apref.imarith(img, op, bias), equal_nan=True)


@pytest.mark.parametrize('N', [8, 64, 72, 128])
def test_fused_division_guards_on_the_sorted_column(ops, apref, N):
    """The reciprocal division's range guards are read off the ends of the sorted column (full AND padded slot counts):
    columns with quotients below 2^-50 or above 2^50, mixed signs, exact zeros and non-finite values must send their wave
    through the exact IEEE path and still match the oracle."""
    rng = np.random.default_rng(900 + N)
    H, W = 8, 192                                            # 6 wavefronts per row group
    bias = rng.normal(100, 2, (H, W)).astype(np.float32)
    dark = rng.normal(10, 1, (H, W)).astype(np.float32)
    nflat = rng.normal(1.0, 0.02, (H, W)).astype(np.float32)
    raw = (rng.normal(600, 20, (N, H, W)) + bias + 0.4 * dark).astype(np.float32)
    e = np.float32(0.4)
    base = (bias + e * dark).astype(np.float32)
    raw[:, 0, 5] = base[0, 5] + np.float32(1e-20)           # quotients ~1e-20 < 2^-50: all one sign
    raw[: N // 2, 1, 70] -= np.float32(1200)                # mixed signs, ordinary magnitudes
    raw[:, 2, 130] = base[2, 130]                           # exact zeros
    raw[3 % N, 2, 130] += np.float32(5)
    raw[:, 3, 10] *= np.float32(1e20)
    nflat[3, 10] = np.float32(1e-25)                        # quotient ~1e47 > 2^50 (and the divisor below 2^-40)
    raw[1 % N, 4, 20] = np.inf
    raw[2 % N, 4, 21] = np.nan
    nflat[5, 100] = np.float32(-1.5)                        # negative flat: the order of the column reverses
    nflat[6, 7] = 0.0                                       # no division at all
    cal = apref.calibrate(raw, bias, dark, nflat, float(e))
    ref = apref.stack_sigclip(cal, sigma=3.0, maxiters=5)
    d = torch.from_numpy(raw).cuda()
    calib = dict(bias=torch.from_numpy(bias).cuda(), dark=torch.from_numpy(dark).cuda(), nflat=torch.from_numpy(nflat).cuda(),
                 exp_ratio=float(e))
    r = ops.stack_sigclip(d, calib=calib, outputs=('mean', 'count'))
    assert np.array_equal(r['count'].cpu().numpy(), ref['count'])
    with np.errstate(over='ignore'):
        assert_ulp(r['mean'].cpu().numpy(), ref['mean'].astype(np.float32), 1, f'guards N={N}')
    med = ops.stack_median(d, calib=calib)
    assert_ulp(med.cpu().numpy(), apref.stack_median(cal).astype(np.float32), 1, f'median guards N={N}')


def test_frames_cut_out_of_an_odd_sized_slab(ops, apref):
    """ApCalibrate.calibrate_files hands cal[i] of an [N, H, W] slab to the per-frame kernels; with H * W % 4 != 0 every
    frame after the first starts off a 16-byte boundary (image arithmetic, threshold mask, bad-pixel repair)."""
    rng = np.random.default_rng(17)
    H, W = 37, 53
    slab = torch.from_numpy(rng.normal(500, 20, (3, H, W)).astype(np.float32)).cuda()
    other = torch.from_numpy(rng.normal(5, 1, (3, H, W)).astype(np.float32)).cuda()
    mask = (rng.random((H, W)) < 0.03).astype(np.uint8)
    for i in range(3):
        assert (slab[i].data_ptr() % 16 != 0) == (i * H * W * 4 % 16 != 0)
        out = ops.imarith(slab[i], 'SUB', other[i])
        assert np.array_equal(out.cpu().numpy(), slab[i].cpu().numpy() - other[i].cpu().numpy())
        m, nbad = ops.threshold_mask(slab[i], 480.0, 520.0)
        ref = ((slab[i].cpu().numpy() < np.float32(480)) | (slab[i].cpu().numpy() > np.float32(520)))
        assert np.array_equal(m.cpu().numpy().astype(bool), ref) and int(nbad.item()) == int(ref.sum())
        fixed, st = ops.fix_badpix(slab[i], torch.from_numpy(mask).cuda(), 2)
        rf, _ = apref.fix_badpix(slab[i].cpu().numpy(), mask, 2)
        assert np.array_equal(fixed.cpu().numpy(), rf, equal_nan=True)


def wide_case(ops, apref, seed):
    """One random stack through stack_sigclip / stack_median against the oracle over the WIDE option space: 1 .. 512 frames,
    sigma down to 0.5, clipping to convergence (the survivors can collapse to a few identical values, or vanish), NaN fractions up
    to 30 %, masters with extreme flats, pedestals, every output plane.  tools/fuzz_long.py runs it on fresh seeds."""
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
        calib = dict(bias=_dev(bias, ops), dark=_dev(dark, ops), nflat=None if nflat is None else _dev(nflat, ops), exp_ratio=_dev(e, ops), pedestal=ped,
                     dark_still_biased=sb)
    else:
        cal, calib = cube.astype(np.float32), None
    pixmask = (rng.random((H, W)) < 0.05).astype(np.uint8) if rng.integers(0, 2) else None
    sigma = float(rng.choice([0.5, 1.5, 2.0, 3.0, 5.0]))
    mi = rng.choice([1, 2, 5, None])
    mi = None if mi is None else int(mi)
    cen = str(rng.choice(['median', 'mean']))
    dv = str(rng.choice(['std', 'std', 'mad_std']))
    outs = [('mean', 'count'), ('mean', 'count', 'median'), ('mean', 'count', 'std'), ('mean', 'count', 'median', 'std'),
            ('mean', 'count', 'mean_f64', 'std_f64'), ('mean', 'count', 'moments_f64'), ('mean', 'count', 'moments')][int(rng.integers(0, 7))]
    sl, su = (sigma, sigma) if rng.integers(0, 3) else (float(rng.choice([0.5, 1.25, 2.0, 3.0, 1e30])), float(rng.choice([0.5, 1.5, 3.0, 4.0, 1e30])))   # not 1.0: two survivors a, b sit ON med -+ 1.0 std
    what = f'seed={seed} N={N} {H}x{W} u16={u16} calib={use_calib} flat={nflat is not None} ped={ped is not None} {cen}/{dv} s={sl}/{su} it={mi} outs={outs}'
    with np.errstate(all='ignore'):
        ref = apref.stack_sigclip(cal, sigma_lower=sl, sigma_upper=su, maxiters=mi, cenfunc=cen, stdfunc=dv, pixmask=pixmask)
    r = ops.stack_sigclip(_dev(cube, ops), sigma_lower=sl, sigma_upper=su, maxiters=mi, cenfunc=cen, stdfunc=dv, calib=calib,
                          pixmask=None if pixmask is None else _dev(pixmask, ops), outputs=outs)
    # Exact ties - a value that EQUALS a bound in exact arithmetic, which few discrete levels (integer frames) and sigma 0.5 / 1.5
    # produce readily (levels L, L+d, L+2d with counts 3, 5, 5: mean - 1.5 std = L) - are decided by the last bit of astropy's
    # sequential float64 sums in frame order, which no sorted-column evaluation reproduces.  A pixel is "on a tie" when the
    # oracle itself changes its answer under a 1e-10 relative change of sigma; those pixels are left out of the comparison.
    with np.errstate(all='ignore'):
        lo_run = apref.stack_sigclip(cal, sigma_lower=sl * (1 - 1e-10), sigma_upper=su * (1 - 1e-10), maxiters=mi, cenfunc=cen, stdfunc=dv,
                                     pixmask=pixmask, want=('count', 'mean'))
        hi_run = apref.stack_sigclip(cal, sigma_lower=sl * (1 + 1e-10), sigma_upper=su * (1 + 1e-10), maxiters=mi, cenfunc=cen, stdfunc=dv,
                                     pixmask=pixmask, want=('count', 'mean'))

    def same(a, b):
        return (a == b) | (np.isnan(a) & np.isnan(b))

    firm = (lo_run['count'] == ref['count']) & (hi_run['count'] == ref['count']) & same(lo_run['mean'], ref['mean']) & same(hi_run['mean'], ref['mean'])
    assert firm.mean() > 0.5, 'too many tie pixels ' + what
    tie = ~firm

    def sel(a, fill=0):
        a = np.array(a, copy=True)
        a[tie] = fill
        return a

    assert np.array_equal(sel(r['count'].cpu().numpy()), sel(ref['count'])), 'count ' + what
    assert_ulp(sel(r['mean'].cpu().numpy()), sel(ref['mean'].astype(np.float32)), 1, 'mean ' + what)
    if 'median' in outs:
        assert_ulp(sel(r['median'].cpu().numpy()), sel(ref['median'].astype(np.float32)), 1, 'median ' + what)
    if 'std' in outs:
        assert_ulp(sel(r['std'].cpu().numpy()), sel(ref['std'].astype(np.float32)), 2, 'std ' + what)
    if 'mean_f64' in outs:
        np.testing.assert_allclose(sel(r['mean_f64'].cpu().numpy()), sel(ref['mean']), rtol=1e-13, atol=1e-11, equal_nan=True, err_msg=what)   # atol: means near zero of values ~1e3
        np.testing.assert_allclose(sel(r['std_f64'].cpu().numpy()), sel(ref['std']), rtol=1e-10, atol=1e-300, equal_nan=True, err_msg=what)
    if 'moments_f64' in outs:
        assert np.array_equal(sel(r['moments_f64']['count'].cpu().numpy()), sel(ref['count'])), 'moments count ' + what
        kept = np.where(ref['keep'], cal.astype(np.float64), 0.0)
        np.testing.assert_allclose(sel(r['moments_f64']['sum'].cpu().numpy()), sel(kept.sum(0)), rtol=1e-12, atol=1e-300, err_msg=what)
    med = ops.stack_median(_dev(cube, ops), calib=calib, pixmask=None if pixmask is None else _dev(pixmask, ops))
    mm = apref.stack_median(cal).astype(np.float32)
    if pixmask is not None:
        mm[pixmask != 0] = np.nan
    assert_ulp(med.cpu().numpy(), mm, 1, 'stack_median ' + what)




@pytest.mark.parametrize('seed', [102910, 103094, 103141, 103148, 103187, 103206, 103314, 103362, 103411, 103430, 103541, 103596, 103625,
                                  103661, 103689, 103717, 103719, 103749, 103791, 103810] + list(range(200000, 200040)))
def test_random_wide_option_space(ops, apref, seed):
    """The first twenty seeds are cases the long fuzz run found (round 2): running moments that had lost their precision after
    the clip collapsed onto near-identical survivors, and survivor sets that vanish."""
    wide_case(ops, apref, seed)
