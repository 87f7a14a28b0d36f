"""F3 + A7 fused (GPU): apgpu_resample_stack_sigclip - resample every frame of an output tile into registers and clip there -
against the oracle's composition apref.stack_sigclip(apref.resample_affine(..)) (survivor counts identical, mean within 1 ulp
of the float64 evaluation) and against this build's own two-step form (same survivors; the values entering the clip are bit for
bit apgpu_resample_affine_f32's).  scripts/resample_all.sh:330-342: one SWarp call resamples and combines."""
import numpy as np
import pytest

from tests.util import assert_ulp

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def ops():
    import torch  # noqa: F401
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


def _frames(rng, N, shape, outliers=True):
    H, W = shape
    yy, xx = np.mgrid[0:H, 0:W]
    sky = 400.0 + 0.05 * xx + 0.03 * yy
    cube = sky[None] + rng.normal(0, 6, (N, H, W))
    if outliers:
        hits = rng.random(cube.shape) < 0.01
        cube[hits] += rng.uniform(100, 4000, hits.sum())
    return cube.astype(np.float32)


def _check(ops, apref, frames, A, mask=None, out_shape=None, sigma=3.0, maxiters=5, cenfunc='median', fscale=None, exact=False,
           conserve_flux=False, mean_tol=1):
    import torch
    fr = torch.from_numpy(frames).cuda()
    mk = None if mask is None else torch.from_numpy(mask).cuda()
    r = ops.resample_stack_sigclip(fr, A, mask=mk, out_shape=out_shape, fscale=fscale, sigma=sigma, maxiters=maxiters, cenfunc=cenfunc,
                                   outputs=('mean', 'count'), exact=exact, conserve_flux=conserve_flux)
    torch.cuda.synchronize()
    res_ref, _ = apref.resample_affine(frames, A, mask=mask, out_shape=out_shape, fscale=fscale, conserve_flux=conserve_flux)
    ref = apref.stack_sigclip(res_ref, sigma=sigma, maxiters=maxiters, cenfunc=cenfunc)
    cnt = r['count'].cpu().numpy()
    assert np.array_equal(cnt, ref['count']), 'survivor counts differ at %s' % (np.argwhere(cnt != ref['count'])[:5].tolist(),)
    frac = assert_ulp(r['mean'].cpu().numpy(), ref['mean'].astype(np.float32), mean_tol, 'fused resample + clip')
    # the two-step form of this build: the same survivors
    res2, _ = ops.resample_affine(fr, A, mask=mk, out_shape=out_shape, fscale=fscale, weight=False, conserve_flux=conserve_flux)
    st2 = ops.stack_sigclip(res2, sigma=sigma, maxiters=maxiters, cenfunc=cenfunc, outputs=('mean', 'count'), exact=exact)
    assert torch.equal(st2['count'], r['count'])
    assert_ulp(r['mean'].cpu().numpy(), st2['mean'].cpu().numpy(), 1, 'fused against two-step')
    return frac, cnt


@pytest.mark.parametrize('N', [16, 13, 12, 8, 5, 3])
def test_fused_matches_oracle_composition(ops, apref, N):
    rng = np.random.default_rng(600 + N)
    frames = _frames(rng, N, (200, 330))
    A = _affines(rng, N)
    A[0] = [1, 0, 0, 0, 1, 0]
    frac, cnt = _check(ops, apref, frames, A)
    assert cnt.max() <= N and cnt[100, 160] >= N - 3


def test_fused_with_bad_pixel_mask_and_nan_inputs(ops, apref):
    rng = np.random.default_rng(611)
    N = 16
    frames = _frames(rng, N, (260, 300))
    frames[3, 40:43, 50:52] = np.nan                              # non-finite input values poison their windows
    frames[7, 200, 10] = np.inf
    mask = (rng.random((260, 300)) < 3e-3).astype(np.uint8)
    mask[120:124, 130:150] = 1                                    # a bad column stub: its footprint exhausts some columns' tails
    A = _affines(rng, N, max_rot_deg=0.3)
    frac, cnt = _check(ops, apref, frames, A, mask=mask)
    assert (cnt < N - 6).any() and (cnt == 0).sum() >= 0
    # every option of the clip: mean centre, one pass, asymmetric sigma through the exact flag
    _check(ops, apref, frames, A, mask=mask, cenfunc='mean', maxiters=2)
    _check(ops, apref, frames, A, mask=mask, exact=True)
    _check(ops, apref, frames, A, mask=mask, sigma=2.0, maxiters=None)


def test_fused_other_output_grid_flux_scale_and_borders(ops, apref):
    rng = np.random.default_rng(612)
    N = 10
    frames = _frames(rng, N, (150, 170))
    A = _affines(rng, N, max_rot_deg=1.5, max_shift=12.0, scale_jitter=0.02)      # wide borders, non-steady tiles
    fs = rng.uniform(0.5, 2.0, N).astype(np.float32)
    _check(ops, apref, frames, A, out_shape=(181, 200), fscale=fs)
    _check(ops, apref, frames, A, out_shape=(64, 64), fscale=fs, conserve_flux=True)


def test_fused_large_transforms_take_the_general_paths(ops, apref):
    rng = np.random.default_rng(613)
    N = 6
    frames = _frames(rng, N, (120, 140), outliers=False)
    A = _affines(rng, N, max_rot_deg=20.0, max_shift=5.0, scale_jitter=0.3)
    A[1] = [3.0, 0.2, -10.0, -0.1, 2.5, 4.0]                      # minification: footprints beyond the staged size
    A[2] = [1e-3, 0, 60.0, 0, 1e-3, 50.0]                         # strong magnification
    _check(ops, apref, frames, A, out_shape=(90, 200))


def test_fused_per_tile_affines(ops, apref):
    import torch
    rng = np.random.default_rng(614)
    N, H, W = 7, 96, 200
    frames = _frames(rng, N, (H, W))
    ty, tx = (H + 15) // 16, (W + 63) // 64
    base = _affines(rng, N)
    A = np.repeat(base[:, None, None, :], ty, 1).repeat(tx, 2).copy()
    A[..., 2] += rng.uniform(-0.05, 0.05, A.shape[:-1])
    A[..., 5] += rng.uniform(-0.05, 0.05, A.shape[:-1])
    mask = (rng.random((H, W)) < 2e-3).astype(np.uint8)
    _check(ops, apref, frames, A, mask=mask)
    _check(ops, apref, frames, A)


def test_fused_moments_feed_the_nshard_combine(ops, apref):
    """The per-GPU share of C5: float64 moments of the fused co-add equal those of the two-step form's survivors."""
    import torch
    rng = np.random.default_rng(615)
    N = 16
    frames = _frames(rng, N, (128, 192))
    A = _affines(rng, N)
    fr = torch.from_numpy(frames).cuda()
    r = ops.resample_stack_sigclip(fr, A, outputs=('moments_f64', 'count'), moments_mean_only=True)
    res2, _ = ops.resample_affine(fr, A, weight=False)
    s2 = ops.stack_sigclip(res2, outputs=('moments_f64', 'count'), moments_mean_only=True)
    torch.cuda.synchronize()
    assert torch.equal(r['count'], s2['count'])
    assert torch.equal(r['moments_f64']['count'], s2['moments_f64']['count'])
    a, b = r['moments_f64']['sum'].cpu().numpy(), s2['moments_f64']['sum'].cpu().numpy()
    np.testing.assert_allclose(a, b, rtol=3e-7, atol=0)


def test_fused_errors_are_loud(ops):
    import torch
    fr = torch.zeros((17, 32, 32), dtype=torch.float32, device='cuda')
    with pytest.raises(ValueError):
        ops.resample_stack_sigclip(fr, np.tile([1, 0, 0, 0, 1, 0], (17, 1)))
    with pytest.raises(ValueError):
        ops.resample_stack_sigclip(fr[:4], np.tile([1, 0, 0, 0, 1, 0], (4, 1)), outputs=('median',))


def test_coadd_fused_option_equals_two_step(ops):
    """ops.coadd(..., fused=True) / ApResample.coadd(fused=True): CLIPPED and AVERAGE through the one-launch kernel - the same
    survivors and (AVERAGE: nothing is clipped, float64 sums on both sides) the same image as the two-step form."""
    import torch
    import astrophotography_amd as ap
    rng = np.random.default_rng(616)
    N = 9
    frames = torch.from_numpy(_frames(rng, N, (130, 150))).cuda()
    A = _affines(rng, N)
    mask = torch.from_numpy((rng.random((130, 150)) < 2e-3).astype(np.uint8)).cuda()
    for combine in ('CLIPPED', 'AVERAGE'):
        a = ops.coadd(frames, A, mask=mask, combine=combine, fused=True)
        b = ops.coadd(frames, A, mask=mask, combine=combine)
        assert torch.equal(a['count'], b['count']), combine
        assert_ulp(a['image'].cpu().numpy(), b['image'].cpu().numpy(), 1, combine)
    r = ap.ApResample('CRITICAL', combine='CLIPPED', conserve_flux=False).coadd(frames, A, mask=mask, fused=True)
    assert torch.equal(r['count'], ops.coadd(frames, A, mask=mask, combine='CLIPPED')['count'])


def test_dithered_sequence_coadds_to_the_scene(ops):
    """End to end, physically: a DITHERED sequence (frame f images the scene warped by the inverse of its registration transform,
    independent noise per frame) resampled with the transforms and clipped gives the scene back - stars included: their columns are
    consistent again, so the clip keeps (nearly) every frame on them, which it cannot on a static scene that the transforms
    misalign (bench.py --c5-dithered).  Both forms of the co-add."""
    import torch
    rng = np.random.default_rng(617)
    N, H, W = 12, 320, 384
    yy, xx = torch.meshgrid(torch.arange(H, device='cuda', dtype=torch.float32), torch.arange(W, device='cuda', dtype=torch.float32), indexing='ij')
    scene = 500.0 + 0.1 * xx
    stars = [(60.3, 80.7, 9000.0), (160.5, 200.2, 20000.0), (250.1, 90.9, 4000.0), (100.0, 300.0, 12000.0)]
    for sy, sx, amp in stars:
        scene = scene + amp * torch.exp(-((xx - sx) ** 2 + (yy - sy) ** 2) / (2 * 2.0 ** 2))
    A = _affines(rng, N, max_rot_deg=0.3, max_shift=3.0)
    frames = torch.empty((N, H, W), dtype=torch.float32, device='cuda')
    g = torch.Generator(device='cuda').manual_seed(617)
    for f in range(N):
        a = A[f]
        Mi = np.linalg.inv(np.array([[a[0], a[1], a[2]], [a[3], a[4], a[5]], [0.0, 0.0, 1.0]]))
        w, _ = ops.resample_affine(scene[None], [[Mi[0, 0], Mi[0, 1], Mi[0, 2], Mi[1, 0], Mi[1, 1], Mi[1, 2]]], weight=False)
        s = torch.nan_to_num(w[0], nan=500.0)
        frames[f] = s + torch.randn((H, W), generator=g, device='cuda') * torch.sqrt(s.clamp_min(1.0))
    inner = (slice(24, H - 24), slice(24, W - 24))                 # away from the borders the warps leave undefined
    for fused in (False, True):
        r = ops.coadd(frames, A, combine='CLIPPED', sigma=3.0, maxiters=5, fused=fused)
        img, cnt = r['image'][inner], r['count'][inner]
        err = (img - scene[inner]).abs()
        noise = torch.sqrt(scene[inner] / N)
        # two Lanczos-3 interpolations of a sigma = 2 px star lose a little of its peak: compare within 4 sigma of the noise + 1.5 %
        assert float((err > 4.0 * noise + 0.015 * scene[inner]).float().mean()) < 2e-3, fused
        for sy, sx, amp in stars:                                 # on the stars' cores the clip keeps (nearly) every frame
            c = cnt[int(sy) - 24 - 1:int(sy) - 24 + 2, int(sx) - 24 - 1:int(sx) - 24 + 2]
            assert int(c.min()) >= N - 2, (fused, sy, sx, c.tolist())
    # the same scene WITHOUT the dither (every frame images it in place) and the same transforms: the resample moves the stars apart,
    # a core's column now runs from core to wing - its spread inflates the clip's sigma, nothing is rejected, and the co-add smears
    # the star (the columns are also what the float32 fast path's guards send to the redo pass: bench.py's redo_fraction 0.086)
    static = scene[None].expand(N, H, W) + torch.randn((N, H, W), generator=g, device='cuda') * torch.sqrt(scene)[None]
    simg = ops.coadd(static.contiguous(), A, combine='CLIPPED', sigma=3.0, maxiters=5)['image'][inner]
    dimg = ops.coadd(frames, A, combine='CLIPPED', sigma=3.0, maxiters=5)['image'][inner]
    for sy, sx, amp in stars:
        iy, ix = int(round(sy)) - 24, int(round(sx)) - 24
        truth = float(scene[inner][iy, ix])
        assert float(dimg[iy, ix]) > 0.95 * truth, (sy, sx, float(dimg[iy, ix]), truth)
        assert float(simg[iy, ix]) < 0.90 * truth, (sy, sx, float(simg[iy, ix]), truth)
