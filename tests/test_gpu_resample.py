"""F3 parity (GPU): apgpu_resample_affine_f32 against the oracle's definition - bit-exact (same table, same
fmaf chains, float64 coordinates) - and the resample + co-add pipeline of config 5 on a small case."""
import numpy as np
import pytest

from tests.util import assert_biteq, assert_ulp

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def ops():
    import torch
    from astrophotography_amd import ops as _ops
    return _ops


@pytest.fixture(scope='module')
def apref():
    from oracle import apref as _a
    return _a


def _affines(rng, n, max_rot_deg=0.2, max_shift=3.0, scale_jitter=0.0):
    out = []
    for _ in range(n):
        th = np.deg2rad(rng.uniform(-max_rot_deg, max_rot_deg))
        s = 1.0 + rng.uniform(-scale_jitter, scale_jitter)
        c, sn = s * np.cos(th), s * np.sin(th)
        out.append([c, -sn, rng.uniform(-max_shift, max_shift), sn, c, rng.uniform(-max_shift, max_shift)])
    return np.array(out, np.float64)


def _run(ops, frames, A, **kw):
    import torch
    mask = kw.pop('mask', None)
    o, w = ops.resample_affine(torch.from_numpy(frames).cuda(), A, mask=None if mask is None else torch.from_numpy(mask).cuda(), **kw)
    torch.cuda.synchronize()
    return o.cpu().numpy(), w.cpu().numpy()


@pytest.mark.parametrize('shape,out_shape', [((96, 130), None), ((70, 67), (81, 75)), ((300, 520), (256, 512))])
def test_resample_bitexact_vs_oracle(ops, apref, shape, out_shape):
    rng = np.random.default_rng(hash(shape) % 1000)
    N = 5
    frames = rng.normal(300, 30, (N,) + shape).astype(np.float32)
    frames[1, 10, 11] = np.nan
    frames[2, 40, 33] = np.inf
    mask = (rng.random(shape) < 0.002).astype(np.uint8) * 3
    A = _affines(rng, N, scale_jitter=0.01)
    A[0] = [1, 0, 0, 0, 1, 0]
    fs = rng.uniform(0.5, 2.0, N).astype(np.float32)
    ref, wref = apref.resample_affine(frames, A, fscale=fs, mask=mask, out_shape=out_shape)
    got, wgot = _run(ops, frames, A, fscale=fs, mask=mask, out_shape=out_shape)
    assert_biteq(got, ref, 'resampled frames')
    assert np.array_equal(wgot, wref)
    assert 0.5 < wref.mean() < 1.0
    # no mask / no fscale / other table resolution
    ref, wref = apref.resample_affine(frames, A, out_shape=out_shape, n_phases=4096)
    got, wgot = _run(ops, frames, A, out_shape=out_shape, n_phases=4096)
    assert_biteq(got, ref, 'resampled frames (no mask, 4096 phases)')
    assert np.array_equal(wgot, wref)


def test_resample_large_transforms_take_the_gather_path(ops, apref):
    """Strong rotation / scale: a tile's input footprint no longer fits LDS; the direct-gather path must give
    the same bits.  Degenerate transforms (all outside, NaN coefficients) give all-NaN frames."""
    rng = np.random.default_rng(7)
    frames = rng.normal(10, 1, (4, 200, 180)).astype(np.float32)
    th = np.deg2rad(33.0)
    A = np.array([[np.cos(th), -np.sin(th), 60.0, np.sin(th), np.cos(th), -20.0],
                  [3.5, 0.0, 5.0, 0.0, 4.0, 5.0],               # 4x minification: 64x16 tile -> 224 x 64 footprint
                  [1.0, 0.0, 1e7, 0.0, 1.0, 0.0],               # entirely outside
                  [np.nan, 0.0, 0.0, 0.0, 1.0, 0.0]])
    ref, wref = apref.resample_affine(frames, A, out_shape=(128, 192))
    got, wgot = _run(ops, frames, A, out_shape=(128, 192))
    assert_biteq(got, ref, 'gather path')
    assert np.array_equal(wgot, wref)
    assert wref[0].mean() > 0.3 and wref[1].mean() > 0.05 and wref[2].sum() == 0 and wref[3].sum() == 0


def test_identity_and_shift_exact(ops):
    rng = np.random.default_rng(3)
    img = rng.normal(100, 10, (64, 200)).astype(np.float32)
    got, w = _run(ops, img, [[1, 0, 5, 0, 1, -0.0 + 3]])
    yy, xx = np.nonzero(w[0])
    assert np.array_equal(got[0][yy, xx], img[yy + 3, xx + 5])


def test_coadd_pipeline_small_config5(ops, apref):
    """Config-5 shape of work on a small case: mask -> per-frame affine resample -> 5-iteration sigma-clipped stack.
    The GPU pipeline must equal 'oracle resample, then oracle stack' (resample bit-exact, mean <= 1 ulp)."""
    import torch
    rng = np.random.default_rng(11)
    N, H, W = 16, 96, 128
    yy, xx = np.mgrid[0:H, 0:W]
    sky = 200 + 0.2 * xx + 40 * np.exp(-((xx - 60) ** 2 + (yy - 50) ** 2) / 30.0)
    frames = (sky[None] + rng.normal(0, 5, (N, H, W))).astype(np.float32)
    hits = rng.random(frames.shape) < 0.003
    frames[hits] += rng.uniform(200, 3000, hits.sum()).astype(np.float32)
    mask = (rng.random((H, W)) < 0.001).astype(np.uint8)
    A = _affines(rng, N)
    fs = np.full(N, 1.0 / 120.0, np.float32)
    res_ref, _ = apref.resample_affine(frames, A, fscale=fs, mask=mask)
    st_ref = apref.stack_sigclip(res_ref, sigma=3.0, maxiters=5)
    r = ops.coadd(torch.from_numpy(frames).cuda(), A, fscale=fs, mask=torch.from_numpy(mask).cuda(), combine='CLIPPED',
                  sigma=3.0, maxiters=5)
    cnt = r['count'].cpu().numpy()
    assert np.array_equal(cnt, st_ref['count'])
    assert_ulp(r['image'].cpu().numpy(), st_ref['mean'].astype(np.float32), 1, 'clipped co-add')
    # MEDIAN / AVERAGE / SUM combine types against numpy on the oracle's resampled frames
    with np.errstate(all='ignore'):
        import warnings
        with warnings.catch_warnings():
            warnings.simplefilter('ignore')
            med = np.nanmedian(res_ref.astype(np.float64), axis=0)
            avg = np.nanmean(res_ref.astype(np.float64), axis=0)
            tot = np.nansum(res_ref.astype(np.float64), axis=0)
    n_ok = np.isfinite(res_ref).sum(0)
    g = ops.coadd(torch.from_numpy(frames).cuda(), A, fscale=fs, mask=torch.from_numpy(mask).cuda(), combine='MEDIAN')
    assert_ulp(g['image'].cpu().numpy(), med.astype(np.float32), 1, 'median co-add')
    g = ops.coadd(torch.from_numpy(frames).cuda(), A, fscale=fs, mask=torch.from_numpy(mask).cuda(), combine='AVERAGE')
    assert np.array_equal(g['count'].cpu().numpy(), n_ok)
    assert_ulp(g['image'].cpu().numpy(), avg.astype(np.float32), 1, 'average co-add')
    g = ops.coadd(torch.from_numpy(frames).cuda(), A, fscale=fs, mask=torch.from_numpy(mask).cuda(), combine='SUM')
    got = g['image'].cpu().numpy()
    ok = n_ok > 0
    assert_ulp(got[ok], tot[ok].astype(np.float32), 1, 'sum co-add')
    with pytest.raises(ValueError):
        ops.coadd(torch.from_numpy(frames).cuda(), A, combine='MODE')


def test_resample_errors_are_loud(ops):
    import torch
    from astrophotography_amd._lib import ApGpuError
    f = torch.zeros((2, 16, 16), device='cuda')
    with pytest.raises(ValueError):
        ops.resample_affine(f, [[1, 0, 0, 0, 1, 0]] * 3)
    with pytest.raises(ValueError):
        ops.resample_affine(f, [[1, 0, 0, 0, 1, 0]] * 2, mask=torch.zeros((8, 8), dtype=torch.uint8, device='cuda'))
    with pytest.raises(ApGpuError):
        ops.resample_affine(torch.zeros((1, 4, 4), device='cuda'), [[1, 0, 0, 0, 1, 0]])


def test_per_tile_affines_bitexact_vs_oracle(ops, apref):
    """One transform per 16 x 64 output tile (the piecewise-affine form of TAN -> TAN): bit for bit against the oracle."""
    from astrophotography_amd import wcs
    rng = np.random.default_rng(19)
    N, H, W = 3, 150, 300
    frames = rng.normal(200, 20, (N, H, W)).astype(np.float32)
    out_shape = (140, 290)
    out_w = wcs.TanWcs.from_center(150.0, 2.2, 1.5, out_shape)
    tiles = []
    for k in range(N):
        th = np.deg2rad(rng.uniform(-3, 3))
        s = (1.5 + rng.uniform(-0.02, 0.02)) / 3600.0
        inp = wcs.TanWcs((W / 2 + rng.uniform(-4, 4), H / 2 + rng.uniform(-4, 4)), (150.0 + rng.uniform(-1e-3, 1e-3), 2.2),
                         [[-s * np.cos(th), s * np.sin(th)], [s * np.sin(th), s * np.cos(th)]])
        tiles.append(wcs.tile_affines(out_w, inp, out_shape))
    tiles = np.stack(tiles, 0)
    mask = (rng.random((H, W)) < 0.002).astype(np.uint8)
    ref, wref = apref.resample_affine(frames, tiles, mask=mask, out_shape=out_shape)
    got, wgot = _run(ops, frames, tiles, mask=mask, out_shape=out_shape)
    assert_biteq(got, ref, 'per-tile affines')
    assert np.array_equal(wgot, wref) and wref.mean() > 0.7
    import torch
    with pytest.raises(ValueError):
        ops.resample_affine(torch.from_numpy(frames).cuda(), tiles[:, :-1], out_shape=out_shape)


def test_wcs_coadd_recovers_star_positions(tmp_path, ops):
    """End to end through files: stars rendered at fixed SKY positions into frames with different TAN WCSs (rotation,
    offset, scale) are co-added by ApResample in WCS mode; their centroids land where the output WCS puts them."""
    import astrophotography_amd as ap
    from astrophotography_amd import fitsio, wcs
    from astrophotography_amd.scripts import ap_coadd
    rng = np.random.default_rng(31)
    N, H, W = 5, 220, 260
    out_shape = (200, 240)
    center = (83.82, -5.39)
    out_w = wcs.TanWcs.from_center(center[0], center[1], 2.0, out_shape)
    sx, sy = rng.uniform(30, 210, 12), rng.uniform(30, 170, 12)          # star positions on the OUTPUT grid
    ra, dec = out_w.pix2sky(sx, sy)
    flux = rng.uniform(3000, 20000, 12)
    names = []
    yy, xx = np.mgrid[0:H, 0:W].astype(np.float64)
    for k in range(N):
        th = np.deg2rad(rng.uniform(-4, 4))
        s = (2.0 * (1 + rng.uniform(-0.01, 0.01))) / 3600.0
        w = wcs.TanWcs((W / 2 + rng.uniform(-6, 6), H / 2 + rng.uniform(-6, 6)), (center[0] + rng.uniform(-2e-3, 2e-3), center[1]),
                       [[-s * np.cos(th), s * np.sin(th)], [s * np.sin(th), s * np.cos(th)]])
        px, py = w.sky2pix(ra, dec)
        img = np.full((H, W), 100.0)
        for x0, y0, f in zip(px, py, flux):
            img += f * np.exp(-((xx - x0) ** 2 + (yy - y0) ** 2) / (2 * 2.2 ** 2))
        img += rng.normal(0, 2.0, (H, W))
        h = fitsio.Header()
        h['EXPTIME'] = 60.0
        for key, val in w.header_cards().items():
            h[key] = val
        fn = tmp_path / f'nav{k}.fits'
        fitsio.write(str(fn), img.astype(np.float32), h)
        names.append(str(fn))
    assert ap_coadd.main([str(tmp_path / 'coadd.fits'), *names, '--combine', 'AVERAGE', '--center', '%f,%f' % center, '--pixelscale', '2.0',
                          '--image_size', '%d,%d' % (out_shape[1], out_shape[0]), '--weight_image', str(tmp_path / 'w.fits'),
                          '-l', 'CRITICAL']) == 0
    co, hdr = fitsio.read(str(tmp_path / 'coadd.fits'))
    wimg, _ = fitsio.read(str(tmp_path / 'w.fits'))
    assert co.shape == out_shape and hdr['CTYPE1'] == 'RA---TAN' and abs(hdr['CRVAL2'] - center[1]) < 1e-12
    assert wimg.max() == N
    oy, ox = np.mgrid[0:out_shape[0], 0:out_shape[1]].astype(np.float64)
    bg = np.nanmedian(co)
    assert abs(bg - 100.0 / 60.0) < 0.05                                 # FSCALE = 1 / EXPTIME
    for x0, y0 in zip(sx, sy):
        sel = ((ox - x0) ** 2 + (oy - y0) ** 2 < 7 ** 2) & np.isfinite(co)
        wgt = np.clip(co[sel] - bg, 0, None)
        cx, cy = (wgt * ox[sel]).sum() / wgt.sum(), (wgt * oy[sel]).sum() / wgt.sum()
        assert abs(cx - x0) < 0.08 and abs(cy - y0) < 0.08, (x0, y0, cx, cy)


def test_flux_conservation(ops, apref):
    """conserve_flux scales by the local pixel-area ratio |det A| (SWarp FSCALASTRO_TYPE VARIABLE): bit-exact against
    the oracle, and the total flux of a star survives a change of pixel scale."""
    rng = np.random.default_rng(41)
    H, W = 160, 200
    yy, xx = np.mgrid[0:H, 0:W].astype(np.float64)
    img = (5000.0 * np.exp(-((xx - 90.3) ** 2 + (yy - 70.6) ** 2) / (2 * 3.0 ** 2))).astype(np.float32)
    A = [[0.8, 0.02, 10.0, -0.02, 0.8, 8.0]]                  # output pixels are 0.8 input pixels wide: finer grid
    ref, _ = apref.resample_affine(img, A, fscale=[0.5], out_shape=(190, 230), conserve_flux=True)
    got, _ = _run(ops, img, A, fscale=np.array([0.5], np.float32), out_shape=(190, 230), conserve_flux=True)
    assert_biteq(got, ref, 'conserve_flux')
    total_in = float(img.astype(np.float64).sum()) * 0.5
    total_out = float(np.nansum(got.astype(np.float64)))
    assert abs(total_out / total_in - 1.0) < 2e-3, (total_in, total_out)
    plain, _ = _run(ops, img, A, fscale=np.array([0.5], np.float32), out_shape=(190, 230))
    det = 0.8 * 0.8 + 0.02 * 0.02
    ok = np.isfinite(plain[0])
    np.testing.assert_allclose(got[0][ok], plain[0][ok] * np.float32(det), rtol=1e-6, atol=1e-6)


def test_oversampling_vs_oracle_composition(ops, apref):
    """SWarp OVERSAMPLING n (resample_all.sh:112, 339): every output pixel = mean of n x n interpolations at its sub-pixel
    centres.  Oracle = the CPU resample on the n-times finer grid (transform of the sub-pixel centres, written out here
    independently of ops.oversampled_affines) followed by a float64 block mean."""
    import torch
    rng = np.random.default_rng(77)
    N, H, W = 3, 90, 120
    out_shape = (70, 100)
    frames = rng.normal(300, 30, (N, H, W)).astype(np.float32)
    frames[1, 30, 40] = np.nan
    A = _affines(rng, N, max_rot_deg=1.0, scale_jitter=0.3)          # output pixels up to 30 % larger than input pixels
    fs = rng.uniform(0.5, 2.0, N).astype(np.float32)
    for n in (2, 4):
        for conserve in (False, True):
            fine = np.empty((N, out_shape[0] * n, out_shape[1] * n), np.float32)
            for i in range(N):
                a0, a1, a2, a3, a4, a5 = A[i]
                off = 0.5 / n - 0.5
                Af = [[a0 / n, a1 / n, a2 + (a0 + a1) * off, a3 / n, a4 / n, a5 + (a3 + a4) * off]]
                scale = np.float32(fs[i] * (n * n if conserve else 1))
                r, _ = apref.resample_affine(frames[i], Af, fscale=[scale], out_shape=fine.shape[1:], conserve_flux=conserve)
                fine[i] = r[0]
            ref = fine.astype(np.float64).reshape(N, out_shape[0], n, out_shape[1], n)
            ref = np.stack([sum(ref[:, :, a, :, b] for a in range(n) for b in range(n))], 0)[0] / (n * n)
            got = ops.resample_oversampled(torch.from_numpy(frames).cuda(), A, n, fscale=fs, out_shape=out_shape, conserve_flux=conserve)
            g = got.cpu().numpy()
            assert np.array_equal(np.isnan(g), np.isnan(ref)), (n, conserve)
            np.testing.assert_allclose(g, ref.astype(np.float32), rtol=2e-7, atol=0, equal_nan=True)
    # oversampling 1 is the plain resample
    one = ops.resample_oversampled(torch.from_numpy(frames).cuda(), A, 1, fscale=fs, out_shape=out_shape).cpu().numpy()
    plain, _ = _run(ops, frames, A, fscale=fs, out_shape=out_shape)
    assert_biteq(one, plain, 'oversampling 1')
    # block mean on its own: row-major float64 accumulation, NaN poisons its block only
    x = rng.normal(0, 1, (12, 20)).astype(np.float32)
    x[5, 7] = np.nan
    bm = ops.block_mean(torch.from_numpy(x).cuda(), 4).cpu().numpy()
    want = np.array([[np.float32(sum(float(x[4 * i + a, 4 * j + b]) for a in range(4) for b in range(4)) / 16.0) for j in range(5)] for i in range(3)])
    assert np.array_equal(bm, want, equal_nan=True) and np.isnan(bm[1, 1]) and np.isfinite(bm).sum() == 14


def test_oversampling_one_pass_bitexact_vs_oracle(ops, apref):
    """apgpu_resample_oversampled_f32 (the n x n sub-samples of a pixel evaluated and averaged in registers) against the
    oracle's one-pass statement, bit for bit: per-frame and per-OUTPUT-tile fine transforms, a bad-pixel mask, rotations from
    registration-sized (the aligned two-copy LDS path) to 25 degrees with a scale change (general / gather path), and the
    two-step form (fine resample + block mean) as a cross-check."""
    import torch
    rng = np.random.default_rng(4242)
    N, H, W = 4, 150, 210
    out_shape = (97, 139)                                                  # ragged last tiles in both directions
    frames = rng.normal(500, 60, (N, H, W)).astype(np.float32)
    frames[2, 70, 100] = np.inf
    mask = (rng.random((H, W)) < 0.002).astype(np.uint8)
    fs = rng.uniform(0.5, 2.0, N).astype(np.float32)
    A = _affines(rng, N, max_rot_deg=0.5, scale_jitter=0.02)
    th = np.deg2rad(25.0)
    A[3] = [1.3 * np.cos(th), -1.3 * np.sin(th), 40.0, 1.3 * np.sin(th), 1.3 * np.cos(th), -20.0]
    t = torch.from_numpy(frames).cuda()
    for n in (2, 3, 4):
        fine_aff, _ = ops.oversampled_affines(A, n, out_shape)
        for conserve in (False, True):
            scale = (fs.astype(np.float64) * (n * n if conserve else 1)).astype(np.float32)
            ref, _ = apref.resample_oversampled(frames, fine_aff.numpy(), n, fscale=scale, mask=mask, out_shape=out_shape, conserve_flux=conserve)
            got = ops.resample_oversampled(t, A, n, fscale=fs, mask=torch.from_numpy(mask).cuda(), out_shape=out_shape, conserve_flux=conserve)
            assert_biteq(got.cpu().numpy(), ref, 'one-pass oversampling %d conserve=%s' % (n, conserve))
            two = ops.resample_oversampled_two_step(t, A, n, fscale=fs, mask=torch.from_numpy(mask).cuda(), out_shape=out_shape, conserve_flux=conserve)
            assert_biteq(two.cpu().numpy(), ref, 'two-step oversampling %d conserve=%s' % (n, conserve))
    assert np.isfinite(ref[:3]).mean() > 0.5 and np.isfinite(ref[3]).any()
    # one transform per OUTPUT tile (wcs.tile_affines(..., tile_scale=n) in ApResample): perturbed copies of the frame's own
    n = 4
    fine_aff, _ = ops.oversampled_affines(A, n, out_shape)
    ty, tx = (out_shape[0] + 15) // 16, (out_shape[1] + 63) // 64
    tiles = np.repeat(np.repeat(fine_aff.numpy()[:, None, None, :], ty, 1), tx, 2)
    tiles[..., 2] += rng.uniform(-0.3, 0.3, tiles.shape[:-1])
    tiles[..., 5] += rng.uniform(-0.3, 0.3, tiles.shape[:-1])
    ref, _ = apref.resample_oversampled(frames, tiles, n, fscale=fs, out_shape=out_shape)
    got = ops.resample_oversampled(t, None, n, fscale=fs, out_shape=out_shape, fine_affines=tiles)
    assert_biteq(got.cpu().numpy(), ref, 'one-pass oversampling, per-tile transforms')
    with pytest.raises(ValueError):
        ops.resample_oversampled(t, None, n, out_shape=out_shape, fine_affines=tiles[:, :-1])


def test_mask_by_scatter_and_inline_agree_with_the_oracle(ops, apref):
    """A bad-pixel mask reaches the output two ways (csrc/resample.hip): few bad pixels are compacted into a list and the
    output pixels whose windows hold them are poisoned after an unmasked resample (mask_scatter_kernel); a list that overflows
    (more than 1/64 of the pixels), per-tile transforms or a strongly magnifying transform apply the mask inside the resample
    kernel.  Every route must give the oracle's NaN pattern and values bit for bit, weight plane included."""
    import torch
    rng = np.random.default_rng(31337)
    N, H, W = 3, 200, 260
    frames = rng.normal(400, 40, (N, H, W)).astype(np.float32)
    A = _affines(rng, N, max_rot_deg=2.0, scale_jitter=0.05)
    A[2] = [0.02, 0.0, 50.0, 0.0, 0.02, 60.0]                               # 50x magnification: the scatter declines this frame
    t = torch.from_numpy(frames).cuda()
    for frac in (0.0005, 0.004, 0.05):                                       # list, list, overflow (> 1/64 of the pixels)
        mask = (rng.random((H, W)) < frac).astype(np.uint8)
        mask[0, 0] = mask[H - 1, W - 1] = mask[100, 130] = 1
        for out_shape in ((H, W), (150, 333)):
            ref, wref = apref.resample_affine(frames, A, mask=mask, out_shape=out_shape)
            got, wgot = ops.resample_affine(t, A, mask=torch.from_numpy(mask).cuda(), out_shape=out_shape)
            assert_biteq(got.cpu().numpy(), ref, 'masked resample, bad fraction %g' % frac)
            assert np.array_equal(wgot.cpu().numpy(), wref)
        n = 3
        fine_aff, _ = ops.oversampled_affines(A, n, (H, W))
        ref, _ = apref.resample_oversampled(frames, fine_aff.numpy(), n, mask=mask, out_shape=(H, W))
        got = ops.resample_oversampled(t, A, n, mask=torch.from_numpy(mask).cuda(), out_shape=(H, W))
        assert_biteq(got.cpu().numpy(), ref, 'masked one-pass oversampling, bad fraction %g' % frac)
    assert np.isnan(ref).mean() > 0.3 and np.isfinite(ref).mean() > 0.05


def test_oversampling_conserves_flux_on_a_coarser_grid(ops):
    """The case oversampling exists for: output pixels 2.5 input pixels wide.  One Lanczos sample per output pixel aliases (the
    flux of a star depends on where it falls); 4 x 4 sub-samples recover the total to a few 1e-3."""
    import torch
    H, W = 200, 240
    yy, xx = np.mgrid[0:H, 0:W].astype(np.float64)
    A = [[2.5, 0.0, 10.0, 0.0, 2.5, 8.0]]
    out_shape = (70, 85)
    err_over, err_single = [], []
    for (cx, cy) in ((121.3, 97.8), (120.0, 98.0), (121.25, 99.25), (122.5, 98.4), (119.1, 96.9), (123.75, 100.5)):
        img = (4000.0 * np.exp(-((xx - cx) ** 2 + (yy - cy) ** 2) / (2 * 1.6 ** 2))).astype(np.float32)
        t = torch.from_numpy(img).cuda()
        total_in = float(img.astype(np.float64).sum())
        over = ops.resample_oversampled(t, A, 4, out_shape=out_shape, conserve_flux=True).cpu().numpy()
        err_over.append(abs(float(np.nansum(over.astype(np.float64))) / total_in - 1.0))
        single, _ = ops.resample_affine(t, A, out_shape=out_shape, weight=False, conserve_flux=True)
        err_single.append(abs(float(np.nansum(single.cpu().numpy().astype(np.float64))) / total_in - 1.0))
    assert max(err_over) < 5e-3, err_over
    assert max(err_single) > 4 * max(err_over), (err_single, err_over)     # what the oversampling is for


def test_weighted_combine(ops, apref):
    """COMBINE_TYPE WEIGHTED with one weight per frame: sum w x / sum w over the finite values, float64, frame order; the weight
    image is the sum of the contributing weights; default weights = 1 / (fscale * clipped std)^2 of every frame."""
    import torch
    rng = np.random.default_rng(78)
    N, H, W = 6, 64, 80
    noise = np.array([2.0, 2.0, 4.0, 8.0, 3.0, 5.0])
    frames = np.stack([rng.normal(200.0, s, (H, W)) for s in noise]).astype(np.float32)
    frames[2, 10:20, 10:30] = np.nan
    frames[:, 40, 41] = np.nan
    w = rng.uniform(0.2, 3.0, N).astype(np.float32)
    mean, wsum = ops.weighted_mean(torch.from_numpy(frames).cuda(), w)
    num, den = np.zeros((H, W)), np.zeros((H, W))
    for i in range(N):
        ok = np.isfinite(frames[i])
        num += np.where(ok, float(w[i]) * frames[i].astype(np.float64), 0.0)
        den += np.where(ok, float(w[i]), 0.0)
    with np.errstate(invalid='ignore', divide='ignore'):
        ref = np.where(den > 0, num / den, np.nan).astype(np.float32)
    assert_biteq(mean.cpu().numpy(), ref, 'weighted mean')
    assert np.array_equal(wsum.cpu().numpy(), den.astype(np.float32)) and wsum[40, 41] == 0 and np.isnan(mean[40, 41].item())
    with pytest.raises(ValueError):
        ops.weighted_mean(torch.from_numpy(frames).cuda(), [1, 1, 0, 1, 1, 1])
    # through coadd: identity registration, weights from the frames' own noise - the noisy frames count less
    A = np.tile([1.0, 0, 0, 0, 1.0, 0], (N, 1))
    fs = np.ones(N, np.float32)
    wts = ops.background_weights(torch.from_numpy(np.nan_to_num(frames, nan=200.0)).cuda(), fs)
    np.testing.assert_allclose(wts, 1.0 / noise ** 2, rtol=0.12)        # 3-sigma clipping reads the noise ~1.5 % low
    r = ops.coadd(torch.from_numpy(frames).cuda(), A, fscale=fs, combine='WEIGHTED', weights=wts)
    img = r['image'].cpu().numpy()
    inner = (slice(8, H - 8), slice(8, W - 8))
    avg = ops.coadd(torch.from_numpy(frames).cuda(), A, fscale=fs, combine='AVERAGE')['image'].cpu().numpy()
    assert np.nanstd(img[inner]) < 0.8 * np.nanstd(avg[inner])
    assert set(r) == {'image', 'count', 'weight'} and int(r['count'].max()) == N
    np.testing.assert_allclose(float(r['weight'][30, 60]), wts.sum(), rtol=1e-6)


def test_coadd_files_oversampled_weighted_wcs(tmp_path, ops):
    """resample_all.sh's wgtavg mode through files: WCS registration on the oversampled grid, weighted mean, weight image =
    sum of weights, GAIN from EGAIN, OVERSAMP card; star centroids land where the output WCS puts them."""
    from astrophotography_amd import fitsio, wcs
    from astrophotography_amd.scripts import ap_coadd
    rng = np.random.default_rng(32)
    N, H, W = 4, 200, 220
    out_shape = (90, 100)
    center = (150.1, 2.2)
    out_w = wcs.TanWcs.from_center(center[0], center[1], 4.0, out_shape)        # output pixels twice the input scale
    sx, sy = rng.uniform(15, 85, 6), rng.uniform(15, 75, 6)
    ra, dec = out_w.pix2sky(sx, sy)
    yy, xx = np.mgrid[0:H, 0:W].astype(np.float64)
    names = []
    for k in range(N):
        th = np.deg2rad(rng.uniform(-3, 3))
        s = 2.0 / 3600.0
        w = wcs.TanWcs((W / 2 + rng.uniform(-4, 4), H / 2 + rng.uniform(-4, 4)), center,
                       [[-s * np.cos(th), s * np.sin(th)], [s * np.sin(th), s * np.cos(th)]])
        px, py = w.sky2pix(ra, dec)
        img = np.full((H, W), 50.0)
        for x0, y0 in zip(px, py):
            img += 9000.0 * np.exp(-((xx - x0) ** 2 + (yy - y0) ** 2) / (2 * 2.5 ** 2))
        img += rng.normal(0, 1.0 + k, (H, W))
        h = fitsio.Header()
        h['EXPOSURE'] = 30.0
        h['EGAIN'] = 1.5
        for key, val in w.header_cards().items():
            h[key] = val
        fn = tmp_path / f'nav{k}.fits'
        fitsio.write(str(fn), img.astype(np.float32), h)
        names.append(str(fn))
    assert ap_coadd.main([str(tmp_path / 'co.fits'), *names, '--combine', 'WEIGHTED', '--oversampling', '4', '--center', '%f,%f' % center,
                          '--pixelscale', '4.0', '--image_size', '%d,%d' % (out_shape[1], out_shape[0]),
                          '--weight_image', str(tmp_path / 'w.fits'), '-l', 'CRITICAL']) == 0
    co, hdr = fitsio.read(str(tmp_path / 'co.fits'))
    wimg, _ = fitsio.read(str(tmp_path / 'w.fits'))
    assert co.shape == out_shape and hdr['OVERSAMP'] == 4 and hdr['COMBINET'] == 'WEIGHTED' and hdr['GAIN'] > 1.5 * 30
    sig = np.array([1.0, 2.0, 3.0, 4.0])
    np.testing.assert_allclose(wimg.max(), np.sum(1.0 / (sig / 30.0) ** 2), rtol=0.1)
    oy, ox = np.mgrid[0:out_shape[0], 0:out_shape[1]].astype(np.float64)
    bg = np.nanmedian(co)
    # flux-conserving resample onto pixels of 4x the area: 4 * 50 / 30 per output pixel
    assert abs(bg - 4 * 50.0 / 30.0) < 0.05
    for x0, y0 in zip(sx, sy):
        sel = ((ox - x0) ** 2 + (oy - y0) ** 2 < 5 ** 2) & np.isfinite(co)
        wgt = np.clip(co[sel] - bg, 0, None)
        cx, cy = (wgt * ox[sel]).sum() / wgt.sum(), (wgt * oy[sel]).sum() / wgt.sum()
        assert abs(cx - x0) < 0.1 and abs(cy - y0) < 0.1, (x0, y0, cx, cy)


def test_steady_tiles_keep_two_weight_rows_bitexact(ops, apref):
    """The fast path's steady tiles (round 5): when the y phase drifts by less than one table row over a lane's consecutive output
    rows, the lane fetches the rows of its first and last output row once and every pixel picks one.  Transforms built to sit on
    the edges of that: unit scale with rotations of a few hundredths of a degree, scale errors just inside and just outside the
    steady band (drift of 0.99 / 1.01 table rows over seven steps), drift in both directions, and offsets that put the y
    fraction across its wrap inside the image - phases .. 1023, 1024 | 0, 1 .., where rows 1024 and 0 are half a phase wide and a
    lane can meet three values.  Frames with NaN / inf and a mask; 1024 and 4096 phases; one transform per frame (32-row
    workgroups) and one per 16 x 64 tile.  Bit for bit against the oracle."""
    rng = np.random.default_rng(77)
    H, W = 300, 520
    out_shape = (256, 512)
    cases = []
    for n_phases in (1024, 4096):
        band = 1.0 / (7 * n_phases)                          # scale error at which the drift over seven steps is one table row
        for eps in (0.0, 3e-5, 0.7 * band, 0.99 * band, 1.01 * band, -0.7 * band, -0.99 * band):
            for th_deg in (0.0, 0.03, -0.11):
                th = np.deg2rad(th_deg)
                s = (1.0 + eps) / max(np.cos(th), 1e-12)     # F4 = s cos(th) = 1 + eps exactly up to rounding
                # y offset: the fraction of Y crosses 1.0 somewhere inside the output rows (and starts just below it)
                ty = 2.0 + (1.0 - abs(eps) * rng.uniform(20, 230) - rng.uniform(0, 1e-4))
                cases.append((n_phases, [s * np.cos(th), -s * np.sin(th), rng.uniform(1, 3), s * np.sin(th), s * np.cos(th), ty]))
    frames = rng.normal(300, 30, (len(cases), H, W)).astype(np.float32)
    frames[:, 100, 200] = np.nan
    frames[:, 17, 300] = np.inf
    mask = (rng.random((H, W)) < 0.002).astype(np.uint8)
    for n_phases in (1024, 4096):
        idx = [i for i, c in enumerate(cases) if c[0] == n_phases]
        A = np.array([cases[i][1] for i in idx], np.float64)
        fr = np.ascontiguousarray(frames[idx])
        ref, wref = apref.resample_affine(fr, A, mask=mask, out_shape=out_shape, n_phases=n_phases)
        got, wgot = _run(ops, fr, A, mask=mask, out_shape=out_shape, n_phases=n_phases)
        assert_biteq(got, ref, 'steady / nearly steady tiles, %d phases' % n_phases)
        assert np.array_equal(wgot, wref)
    # one transform per 16 x 64 tile (16-row workgroups): the same frames' transforms, perturbed per tile by a few 1e-6
    gy, gx = (out_shape[0] + 15) // 16, (out_shape[1] + 63) // 64
    idx = [i for i, c in enumerate(cases) if c[0] == 1024][:6]
    tiles = np.empty((len(idx), gy, gx, 6), np.float64)
    for k, i in enumerate(idx):
        a = np.array(cases[i][1])
        for ty_ in range(gy):
            for tx_ in range(gx):
                p = a.copy()
                p[[0, 4]] += rng.uniform(-2e-6, 2e-6)
                tiles[k, ty_, tx_] = p
    fr = np.ascontiguousarray(frames[idx])
    ref, wref = apref.resample_affine(fr, tiles, mask=mask, out_shape=out_shape)
    got, wgot = _run(ops, fr, tiles, mask=mask, out_shape=out_shape)
    assert_biteq(got, ref, 'steady tiles, one transform per tile')
    assert np.array_equal(wgot, wref)
