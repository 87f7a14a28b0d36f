#!/usr/bin/env python3
"""Time-budgeted randomised parity run (development aid): random stacks (1 .. 512 frames, float32 / uint16, fused calibration
with awkward masters, pixel masks, every clip option and output plane) and the test suite's own random cases with fresh seeds,
HIP kernels against the oracle.  Prints every mismatch and a summary.

    python tools/fuzz_long.py --minutes 8 [--seed0 100000]
"""
import argparse
import os
import sys
import time
import traceback

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from astrophotography_amd import ops
from oracle import apref
from tests import test_gpu_fuzz as tf
from tests.util import assert_ulp, synth_cube, synth_masters


def dev(a):
    a = np.ascontiguousarray(a)
    return ops.to_device_u16(a) if a.dtype == np.uint16 else torch.from_numpy(a).cuda()


def big_case(seed):
    return tf.wide_case(ops, apref, seed)


def global_case(seed):
    """ops.sigclip_global (float32 / float64 / uint16) against the oracle over awkward populations: few discrete levels,
    constants, heavy tails, NaN / inf, sizes around the 8192-element piece and tile edges, sigma from 0.5, any iteration count."""
    rng = np.random.default_rng(seed)
    n = int(rng.choice([int(rng.integers(1, 40)), int(rng.integers(40, 9000)), int(rng.integers(8000, 70000)), 8192 * int(rng.integers(1, 6)),
                        2048 * int(rng.integers(1, 9)) + int(rng.integers(-2, 3))]))
    kind = int(rng.integers(0, 5))
    if kind == 0:
        x = rng.normal(20, 3, n)
    elif kind == 1:
        x = rng.choice(rng.normal(100, 5, int(rng.integers(1, 5))), n)                 # a few discrete levels
    elif kind == 2:
        x = np.full(n, rng.normal(0, 100))
    elif kind == 3:
        x = rng.standard_cauchy(n) * 10 + 500
    else:
        x = np.rint(rng.normal(1000, 8, n))
    dt = [np.float32, np.float64, np.uint16][int(rng.integers(0, 3))]
    if dt == np.uint16:
        x = np.clip(np.rint(np.abs(x)), 0, 65535).astype(np.uint16)
    else:
        x = x.astype(dt)
        if n > 10 and rng.integers(0, 2):
            x[rng.integers(0, n, 3)] = np.nan
            x[rng.integers(0, n, 2)] = np.inf * rng.choice([-1, 1])
    sigma = float(rng.choice([0.5, 1.0, 1.5, 2.0, 3.0, 4.0, 5.0]))
    mi = rng.choice([1, 2, 5, 10, None])
    mi = None if mi is None else int(mi)
    what = f'global seed={seed} n={n} kind={kind} {np.dtype(dt).name} s={sigma} it={mi}'
    with np.errstate(all='ignore'):
        ref = apref.sigclip_global(x, sigma=sigma, maxiters=mi)
    t = ops.to_device_u16(x) if dt == np.uint16 else torch.from_numpy(x).cuda()
    st = ops.sigclip_global(t, sigma=sigma, maxiters=mi).cpu().numpy()
    got = [st[0], st[1], st[2]]
    want = [ref['mean'], ref['median'], ref['std']]
    if dt == np.float32:
        got, want = [np.float32(v) for v in got], [np.float32(v) for v in want]
    same = all((g == w) or (g != g and w != w) for g, w in zip(got, want))
    assert same, f'{what}: {got} vs {want}'
    assert int(st[6]) == ref['nkeep'] and int(st[5]) == ref['niter'], f'{what}: kept {st[6]} iter {st[5]} vs {ref["nkeep"]} {ref["niter"]}'


def calibrate_case(seed):
    """Mixed-precision calibrate + flat normalisation + bad-pixel repair (any dtype mix, any deltapix) against the oracle."""
    rng = np.random.default_rng(seed)
    H, W = int(rng.integers(1, 40)), int(rng.integers(1, 200))
    N = int(rng.integers(1, 5))
    bias, dark, flat = synth_masters(rng, (H, W))
    if W > 6 and rng.integers(0, 2):
        flat[0, :6] = [0.0, np.nan, -1.25, np.inf, 1e-30, 2.0]
    mdt = [np.float32, np.float64]
    bias = bias.astype(mdt[int(rng.integers(0, 2))])
    dark = dark.astype(mdt[int(rng.integers(0, 2))])
    flat = flat.astype(mdt[int(rng.integers(0, 2))])
    rdt = [np.uint16, np.float32, np.float64][int(rng.integers(0, 3))]
    raw = synth_cube(rng, N, (H, W), dtype=np.uint16 if rdt == np.uint16 else np.float32).astype(rdt)
    use_flat = bool(rng.integers(0, 3))
    nflat_ref = apref.flat_normalize(flat)[0] if use_flat else None
    e = rng.uniform(0.2, 2.0, N)
    ped = np.where(rng.random(N) < 0.4, rng.uniform(-100, 100, N), 0.0) if rng.integers(0, 2) else None
    sb = bool(rng.integers(0, 2))
    what = f'calibrate seed={seed} {N}x{H}x{W} raw={np.dtype(rdt).name} bias={bias.dtype} dark={dark.dtype} flat={flat.dtype if use_flat else None} ped={ped is not None} sb={sb}'
    ref = apref.calibrate_mixed(raw, bias, dark, nflat_ref, e, ped, sb)
    nflat = None
    if use_flat:
        nflat, _ = ops.flat_normalize(dev(flat))
        assert np.array_equal(nflat.cpu().numpy(), nflat_ref, equal_nan=True), 'nflat ' + what
    out = ops.calibrate(dev(raw), dev(bias), dev(dark), nflat, e, pedestal=ped, dark_still_biased=sb)
    assert out.cpu().numpy().dtype == ref.dtype and np.array_equal(out.cpu().numpy(), ref, equal_nan=True), what
    mask = (rng.random((H, W)) < float(rng.choice([0.0, 0.02, 0.3]))).astype(np.uint8)
    delta = int(rng.integers(1, 6))
    img = ref[0]
    fr, _ = apref.fix_badpix(img, mask, delta)
    fg, _ = ops.fix_badpix(out[0].contiguous(), dev(mask), delta)
    assert np.array_equal(fg.cpu().numpy(), fr, equal_nan=True), f'fix_badpix delta={delta} ' + what


def frame_case(seed):
    """The F4 kernels against their restatements on small random fields: box statistics (any box size, both LDS-resident and
    streamed boxes), source mask, L.A.Cosmic (both fine-structure modes)."""
    from oracle import background_ref as br
    from oracle import lacosmic_ref as lr
    rng = np.random.default_rng(seed)
    H, W = int(rng.integers(20, 120)), int(rng.integers(20, 160))
    yy, xx = np.mgrid[0:H, 0:W]
    img = rng.normal(400.0, 8.0, (H, W))
    for _ in range(int(rng.integers(0, 12))):
        cy, cx, amp = rng.uniform(0, H), rng.uniform(0, W), rng.uniform(100, 8000)
        img += amp * np.exp(-((xx - cx) ** 2 + (yy - cy) ** 2) / (2 * rng.uniform(1.0, 2.5) ** 2))
    if rng.integers(0, 2):
        img = np.rint(img)                                   # integer-valued: ties in the box medians
    img = img.astype(np.float32)
    if rng.integers(0, 2):
        img[rng.integers(0, H), rng.integers(0, W)] = np.nan
    mask = (rng.random((H, W)) < float(rng.choice([0.0, 0.03, 0.4]))).astype(np.uint8)
    bh, bw = int(rng.integers(3, H + 10)), int(rng.integers(3, W + 10))
    if rng.integers(0, 3) == 0:
        bh, bw = H, W                                        # one box: the streamed (non-resident) path for larger fields
    sigma = float(rng.choice([1.5, 2.0, 3.0]))
    mi = int(rng.choice([1, 5, 10]))
    what = f'frame seed={seed} {H}x{W} boxes {bh}x{bw} s={sigma} it={mi}'
    st = ops.box_clipped_stats(torch.from_numpy(img).cuda(), torch.from_numpy(mask).cuda(), bh, bw, sigma=sigma, maxiters=mi).cpu().numpy()
    med, std, nfin, nm0 = br.box_clipped_stats(img, mask, bh, bw, sigma, mi)
    lo = br.box_clipped_stats(img, mask, bh, bw, sigma * (1 - 1e-10), mi)[2]
    hi = br.box_clipped_stats(img, mask, bh, bw, sigma * (1 + 1e-10), mi)[2]
    firm = (lo == nfin) & (hi == nfin)                       # boxes on an exact tie are left out (see DESIGN)
    assert np.array_equal(st[..., 2].astype(np.int64)[firm], nfin[firm]), 'box survivors ' + what
    assert np.array_equal(st[..., 3].astype(np.int64), nm0), 'box masked ' + what
    assert np.array_equal(st[..., 0][firm], med[firm], equal_nan=True), 'box median ' + what
    np.testing.assert_allclose(st[..., 1][firm], std[firm], rtol=1e-11, equal_nan=True, err_msg=what)
    # cosmic rays
    clean_img = np.nan_to_num(img, nan=400.0)
    for _ in range(int(rng.integers(0, 30))):
        clean_img[rng.integers(3, H - 3), rng.integers(3, W - 3)] += rng.uniform(300, 6000)
    gain = float(rng.choice([1.0, 1.3]))
    fsmode = str(rng.choice(['convolve', 'median']))
    rc, rm = lr.detect_cosmics(clean_img, gain=gain, satlevel=gain * 65535, fsmode=fsmode)
    e = torch.from_numpy(clean_img).cuda() * np.float32(gain)
    gc, gm, _ = ops.lacosmic(e, satlevel=gain * 65535, fsmode=fsmode)
    assert np.array_equal(gm.cpu().numpy().astype(bool), rm), f'lacosmic mask {fsmode} ' + what
    assert np.array_equal(gc.cpu().numpy(), rc, equal_nan=True), f'lacosmic image {fsmode} ' + what


def resample_case(seed):
    """apgpu_resample_affine_f32 against the oracle: any rotation / scale / shift (LDS-staged and direct-gather tiles), masks,
    flux scales, output shapes, table resolutions."""
    rng = np.random.default_rng(seed)
    big = rng.integers(0, 3) == 0                            # room for interior tiles: the aligned two-copy LDS path
    N, H, W = int(rng.integers(1, 4)), int(rng.integers(6, 400 if big else 150)), int(rng.integers(6, 600 if big else 200))
    frames = rng.normal(300, 30, (N, H, W)).astype(np.float32)
    if rng.integers(0, 2):
        frames[rng.integers(0, N), rng.integers(0, H), rng.integers(0, W)] = np.nan
    A = []
    for _ in range(N):
        th = np.deg2rad(rng.uniform(-180, 180) if rng.integers(0, 4) == 0 else rng.uniform(-1, 1))
        sc = float(rng.choice([1.0, rng.uniform(0.3, 3.0)]))
        c, sn = sc * np.cos(th), sc * np.sin(th)
        A.append([c, -sn, rng.uniform(-20, 20), sn, c, rng.uniform(-20, 20)])
    A = np.array(A)
    out_shape = None if rng.integers(0, 2) else (int(rng.integers(1, 180)), int(rng.integers(1, 260)))
    mask = (rng.random((H, W)) < 0.01).astype(np.uint8) if rng.integers(0, 2) else None
    fs = rng.uniform(0.1, 3.0, N).astype(np.float32) if rng.integers(0, 2) else None
    nph = int(rng.choice([64, 1024, 4096]))
    cf = bool(rng.integers(0, 2))
    what = f'resample seed={seed} {N}x{H}x{W} -> {out_shape} phases={nph} conserve={cf}'
    ref, wref = apref.resample_affine(frames, A, fscale=fs, mask=mask, out_shape=out_shape, n_phases=nph, conserve_flux=cf)
    got, wgot = ops.resample_affine(torch.from_numpy(frames).cuda(), A, fscale=fs, mask=None if mask is None else torch.from_numpy(mask).cuda(),
                                    out_shape=out_shape, n_phases=nph, conserve_flux=cf)
    assert np.array_equal(got.cpu().numpy(), ref, equal_nan=True), what
    assert np.array_equal(wgot.cpu().numpy(), wref), 'weights ' + what
    # OVERSAMPLING in one pass against the oracle's one-pass statement (per-frame or per-output-tile fine transforms)
    n = int(rng.integers(2, 6))
    h, w = (H, W) if out_shape is None else out_shape
    if h * w * n * n * N < 3_000_000:
        fine_aff, _ = ops.oversampled_affines(A, n, (h, w))
        fine_aff = fine_aff.numpy()
        fs_n = np.ones(N, np.float32) if fs is None else fs
        scale = (fs_n.astype(np.float64) * (n * n if cf else 1)).astype(np.float32)
        mk = None if mask is None else torch.from_numpy(mask).cuda()
        if rng.integers(0, 2):
            ref2, _ = apref.resample_oversampled(frames, fine_aff, n, fscale=scale, mask=mask, out_shape=(h, w), n_phases=nph, conserve_flux=cf)
            got2 = ops.resample_oversampled(torch.from_numpy(frames).cuda(), A, n, fscale=fs_n, mask=mk, out_shape=(h, w), n_phases=nph, conserve_flux=cf)
        else:
            ty, tx = (h + 15) // 16, (w + 63) // 64
            tiles = np.repeat(np.repeat(fine_aff[:, None, None, :], ty, 1), tx, 2)
            tiles[..., 2] += rng.uniform(-0.4, 0.4, tiles.shape[:-1])
            tiles[..., 5] += rng.uniform(-0.4, 0.4, tiles.shape[:-1])
            ref2, _ = apref.resample_oversampled(frames, tiles, n, fscale=scale, mask=mask, out_shape=(h, w), n_phases=nph, conserve_flux=cf)
            got2 = ops.resample_oversampled(torch.from_numpy(frames).cuda(), None, n, fscale=fs_n, mask=mk, out_shape=(h, w), n_phases=nph,
                                            conserve_flux=cf, fine_affines=tiles)
        assert np.array_equal(got2.cpu().numpy(), ref2, equal_nan=True), f'oversampling {n} ' + what


def arith_case(seed):
    """ApImArith.apply over dtype mixes against NumPy itself: the reference computes ufunc(data1, data2, out=zeros_like(data1))
    (core/ApImArith.py:210-232), so values, wrap-around and the operand combinations NumPy refuses all follow from NumPy."""
    from astrophotography_amd.core.ApImArith import ApImArith
    rng = np.random.default_rng(seed)
    H, W = int(rng.integers(1, 30)), int(rng.integers(1, 70))
    dt1 = [np.uint16, np.float32, np.float64][int(rng.integers(0, 3))]
    a = (rng.integers(0, 65536, (H, W)) if dt1 == np.uint16 else rng.normal(100, 50, (H, W))).astype(dt1)
    if dt1 != np.uint16 and rng.integers(0, 2):
        a[rng.integers(0, H), rng.integers(0, W)] = [np.nan, np.inf, 0.0][int(rng.integers(0, 3))]
    if rng.integers(0, 3) == 0:
        b = float(rng.choice([0.0, 2.5, -3.0, 1e30]))
    else:
        dt2 = [np.uint16, np.int16, np.float32, np.float64, dt1, dt1][int(rng.integers(0, 6))]
        b = (rng.integers(0, 3000, (H, W)) if np.dtype(dt2).kind in 'ui' else rng.normal(5, 2, (H, W))).astype(dt2)
        if np.dtype(dt2).kind == 'f' and rng.integers(0, 2):
            b[rng.integers(0, H), rng.integers(0, W)] = 0.0
    op = ['ADD', 'SUB', 'MUL', 'DIV'][int(rng.integers(0, 4))]
    uf = {'ADD': np.add, 'SUB': np.subtract, 'MUL': np.multiply, 'DIV': np.divide}[op]
    what = f'arith seed={seed} {np.dtype(dt1).name} {op} {b if isinstance(b, float) else b.dtype.name}'
    want, err = None, None
    with np.errstate(all='ignore'):
        try:
            want = np.zeros_like(a)
            uf(a, b, out=want)
        except TypeError as ex:                              # UFuncTypeError: casting refused
            err = ex
    try:
        got = ApImArith('CRITICAL').apply(a, op, b)
    except TypeError:
        assert err is not None, 'GPU path raised TypeError, NumPy did not: ' + what
        return
    assert err is None, f'NumPy refuses ({err}), GPU path did not: ' + what
    assert got.dtype == want.dtype and np.array_equal(got, want, equal_nan=True), what


def fits_case(seed):
    """FITS round trips: host write -> device read (every BITPIX, the uint16 BZERO convention, odd shapes) equals the host
    read; device write of float32 / float64 -> host read gives the tensor back bit for bit, headers included."""
    import tempfile
    from astrophotography_amd import fitsio
    rng = np.random.default_rng(seed)
    H, W = int(rng.integers(1, 60)), int(rng.integers(1, 90))
    dt = [np.uint8, np.int16, np.uint16, np.int32, np.float32, np.float64][int(rng.integers(0, 6))]
    if np.dtype(dt).kind in 'ui':
        info = np.iinfo(dt)
        a = rng.integers(info.min, int(info.max) + 1, (H, W)).astype(dt)
    else:
        a = rng.normal(0, 1e3, (H, W)).astype(dt)
        a[rng.integers(0, H), rng.integers(0, W)] = [np.nan, np.inf, -0.0][int(rng.integers(0, 3))]
    hdr = fitsio.Header()
    hdr['EXPTIME'] = float(rng.uniform(0.1, 600))
    hdr['OBJECT'] = 'x' * int(rng.integers(0, 120))
    what = f'fits seed={seed} {np.dtype(dt).name} {H}x{W}'
    with tempfile.TemporaryDirectory() as d:
        f = os.path.join(d, 'a.fits')
        fitsio.write(f, a, hdr)
        back, h2 = fitsio.read(f)
        assert back.dtype == a.dtype and np.array_equal(back, a, equal_nan=True), 'host round trip ' + what
        assert h2['OBJECT'] == hdr['OBJECT'] and h2['EXPTIME'] == hdr['EXPTIME'], 'header ' + what
        t, h3 = fitsio.read_device(f)
        got = t.view(torch.int16).cpu().numpy().view(np.uint16) if t.dtype == torch.uint16 else t.cpu().numpy()
        want = a if a.dtype in (np.uint16, np.float32, np.float64) else a.astype(got.dtype)
        assert np.array_equal(got, want, equal_nan=True), f'device read ({got.dtype}) ' + what
        if a.dtype in (np.float32, np.float64):
            g = os.path.join(d, 'b.fits')
            fitsio.write_device(g, torch.from_numpy(a).cuda(), h2)
            again, h4 = fitsio.read(g)
            assert again.dtype == a.dtype and np.array_equal(again.view(np.uint8), a.view(np.uint8)), 'device write ' + what
            assert h4['OBJECT'] == hdr['OBJECT'], 'device write header ' + what


def chunked_case(seed):
    """More than 512 frames on one GPU (ops.stack_sigclip_chunked / ApStack): an unclipped mean is exact to the float64 sum; a
    clipped one equals "oracle per chunk, moments added" (the hierarchical semantics of the N-shard combine)."""
    rng = np.random.default_rng(seed)
    N = int(rng.integers(513, 1400))
    H, W = int(rng.integers(1, 4)), int(rng.integers(1, 90))
    cube = synth_cube(rng, N, (H, W), nan_frac=float(rng.choice([0.0, 0.05])))
    chunk = int(rng.choice([0, int(rng.integers(64, 513))]))
    sigma = float(rng.choice([2.0, 3.0, 1e30]))
    mi = int(rng.choice([1, 5]))
    what = f'chunked seed={seed} N={N} {H}x{W} chunk={chunk} s={sigma} it={mi}'
    r = ops.stack_sigclip_chunked(torch.from_numpy(cube).cuda(), chunk=chunk or None, want_std=True, sigma=sigma, maxiters=mi)
    if chunk == 0:
        parts = -(-N // 512)
        chunk = -(-N // parts)
    tot, cnt, sq = np.zeros((H, W)), np.zeros((H, W), np.int64), np.zeros((H, W))
    for i0 in range(0, N, chunk):
        part = cube[i0:i0 + chunk]
        with np.errstate(all='ignore'):
            ref = apref.stack_sigclip(part, sigma=sigma, maxiters=mi)
        kept = np.where(ref['keep'], part.astype(np.float64), 0.0)
        tot += kept.sum(0)
        sq += (kept * kept).sum(0)
        cnt += ref['count']
    assert np.array_equal(r['count'].cpu().numpy(), cnt), 'count ' + what
    with np.errstate(all='ignore'):
        mean = (tot / cnt).astype(np.float32)
    assert_ulp(r['mean'].cpu().numpy(), mean, 1, 'mean ' + what)
    with np.errstate(all='ignore'):
        var = sq / cnt - (tot / cnt) ** 2
    got = r['std'].cpu().numpy().astype(np.float64)
    ok = cnt > 0
    np.testing.assert_allclose(got[ok] ** 2, np.maximum(var[ok], 0), rtol=1e-4, atol=1e-3, err_msg=what)


def a6_case(seed):
    """The ccdproc.combine configuration (one pass of median / mad_std, float64 planes) on its fast kernel + rich kernel pair
    (stack_mad.hip; a quarter of the cases 129 .. 512 frames: the chunked passes of stack_chunks.hip): float32 / uint16, thresholds, noise levels from a few distinct integers (ties, MAD = 0) to
    wide, outliers on one or both sides, NaN / inf, constant columns, any image size - against the oracle's restatement."""
    rng = np.random.default_rng(seed)
    N = int(rng.integers(3, 129))
    H, W = int(rng.integers(1, 6)), int(rng.integers(1, 400))
    if rng.integers(0, 4) == 0:                               # round 6: 129 .. 512 frames on the chunked order-statistics passes
        N = int(rng.choice([int(rng.integers(129, 513)), int(rng.choice([129, 192, 193, 256, 257, 384, 385, 512]))]))
        W = int(rng.integers(1, 130))
    u16 = bool(rng.integers(0, 2))
    sig = float(rng.choice([0.3, 1.0, 3.0, 40.0]))
    cube = rng.normal(float(rng.choice([50.0, 1000.0, 30000.0])), sig, (N, H, W))
    if rng.integers(0, 2):
        cube = np.rint(cube)
    frac = float(rng.choice([0.0, 0.01, 0.1, 0.3]))
    hits = rng.random(cube.shape) < frac
    cube[hits] += rng.uniform(-1, 1, hits.sum()) * float(rng.choice([5.0, 100.0, 5000.0])) * sig
    if W > 3:
        cube[:, :, 0] = cube[0, 0, 0]
        k = int(rng.integers(1, N))
        cube[:k, :, 1] += 50 * sig
    lo, hi = float(rng.choice([1.0, 3.0, 5.0])), float(rng.choice([2.0, 5.0]))
    form = str(rng.choice(['legacy', 'astropy']))            # which published Combiner.sigma_clipping (golden group G12 holds both)
    if u16:
        cube = np.clip(np.rint(cube), 0, 65535).astype(np.uint16)
        ref = apref.combine_ccdproc(cube.astype(np.float32), lo, hi, form=form)
    else:
        cube = cube.astype(np.float32)
        bad = rng.random(cube.shape) < float(rng.choice([0.0, 0.0, 0.002, 0.05]))
        cube[bad] = rng.choice(np.array([np.nan, np.inf, -np.inf], np.float32), bad.sum())
        with np.errstate(all='ignore'):
            ref = apref.combine_ccdproc(cube, lo, hi, form=form)
    what = f'a6 seed={seed} N={N} {H}x{W} u16={u16} lo={lo} hi={hi} form={form}'
    r = ops.stack_sigclip(dev(cube), sigma_lower=lo, sigma_upper=hi, maxiters=1, cenfunc='median', stdfunc='mad_std',
                          outputs=('mean', 'count', 'mean_f64', 'std_f64'), nonfinite_unclipped=(form == 'astropy'))
    assert np.array_equal(r['count'].cpu().numpy(), ref['count']), 'count ' + what
    # (the oracle sums in frame order: its own rounding is up to n eps max|x|, which is all there is to a mean near zero)
    fin = np.abs(cube[np.isfinite(cube)].astype(np.float64))
    scale = float(fin.max()) if fin.size else 1.0
    np.testing.assert_allclose(r['mean_f64'].cpu().numpy(), ref['mean'], rtol=4e-16, atol=1e-14 * scale, equal_nan=True, err_msg=what)
    np.testing.assert_allclose(r['std_f64'].cpu().numpy(), ref['std'], rtol=1e-12, atol=1e-12 * scale, equal_nan=True, err_msg=what)


def fused_case(seed):
    """apgpu_resample_stack_sigclip (resample + clip in one launch) against the oracle's composition stack_sigclip(resample_affine):
    1 .. 16 frames, registration-sized and large transforms (fast, staged and gather tiles; frame borders), masks (by hit bits and
    at the fill), non-finite inputs, flux scales, output shapes, per-tile transforms, every clip option, exact flag."""
    rng = np.random.default_rng(seed)
    N = int(rng.integers(1, 17))
    big = rng.integers(0, 3) == 0
    H, W = int(rng.integers(6, 300 if big else 120)), int(rng.integers(6, 500 if big else 200))
    frames = (rng.normal(float(rng.choice([0.0, 300.0, 20000.0])), float(rng.choice([1.0, 30.0])), (N, H, W))).astype(np.float32)
    hits = rng.random(frames.shape) < float(rng.choice([0.0, 0.01, 0.05]))
    frames[hits] += rng.uniform(100, 5000, hits.sum()).astype(np.float32)
    if rng.integers(0, 2):
        bad = rng.random(frames.shape) < 1e-3
        frames[bad] = rng.choice(np.array([np.nan, np.inf, -np.inf], np.float32), bad.sum())
    wild = rng.integers(0, 4) == 0
    A = []
    for _ in range(N):
        th = np.deg2rad(rng.uniform(-180, 180) if wild and rng.integers(0, 2) else rng.uniform(-1, 1))
        sc = float(rng.choice([1.0, rng.uniform(0.5, 2.0)])) if wild else 1.0 + rng.uniform(-1e-3, 1e-3)
        c, sn = sc * np.cos(th), sc * np.sin(th)
        A.append([c, -sn, rng.uniform(-8, 8), sn, c, rng.uniform(-8, 8)])
    A = np.array(A)
    out_shape = None if rng.integers(0, 2) else (int(rng.integers(1, 150)), int(rng.integers(1, 260)))
    h, w = (H, W) if out_shape is None else out_shape
    if rng.integers(0, 4) == 0:                              # one transform per 16 x 64 output tile
        ty, tx = (h + 15) // 16, (w + 63) // 64
        A = np.repeat(np.repeat(A[:, None, None, :], ty, 1), tx, 2).copy()
        A[..., 2] += rng.uniform(-0.2, 0.2, A.shape[:-1])
        A[..., 5] += rng.uniform(-0.2, 0.2, A.shape[:-1])
    mask = (rng.random((H, W)) < float(rng.choice([1e-3, 0.02, 0.2]))).astype(np.uint8) if rng.integers(0, 2) else None
    fs = rng.uniform(0.1, 3.0, N).astype(np.float32) if rng.integers(0, 2) else None
    nph = int(rng.choice([64, 1024, 4096]))                  # 4096: the weight table does not fit LDS - the first kernel form
    cf = bool(rng.integers(0, 2))
    sigma = float(rng.choice([1.5, 2.0, 3.0, 5.0]))         # (not 1.0: two survivors a, b sit ON centre -+ 1.0 std - exact ties)
    maxiters = [1, 2, 5, None][rng.integers(0, 4)]
    cen = str(rng.choice(['median', 'mean']))
    exact = bool(rng.integers(0, 3) == 0)
    what = f'fused seed={seed} {N}x{H}x{W} -> {out_shape} phases={nph} sigma={sigma} maxiters={maxiters} cen={cen} exact={exact} per_tile={A.ndim == 4}'
    with np.errstate(all='ignore'):
        res_ref, _ = apref.resample_affine(frames, A, fscale=fs, mask=mask, out_shape=out_shape, n_phases=nph, conserve_flux=cf)
        ref = apref.stack_sigclip(res_ref, sigma=sigma, maxiters=maxiters, cenfunc=cen)
    r = ops.resample_stack_sigclip(torch.from_numpy(frames).cuda(), A, fscale=fs, mask=None if mask is None else torch.from_numpy(mask).cuda(),
                                   out_shape=out_shape, n_phases=nph, conserve_flux=cf, sigma=sigma, maxiters=maxiters, cenfunc=cen,
                                   outputs=('mean', 'count'), exact=exact)
    # exact ties (a value that equals a bound in exact arithmetic): pixels on which the oracle itself changes its answer under a
    # 1e-10 relative change of sigma are left out, as in tests/test_gpu_fuzz.py wide_case
    with np.errstate(all='ignore'):
        lo_run = apref.stack_sigclip(res_ref, sigma=sigma * (1 - 1e-10), maxiters=maxiters, cenfunc=cen, want=('count', 'mean'))
        hi_run = apref.stack_sigclip(res_ref, sigma=sigma * (1 + 1e-10), maxiters=maxiters, cenfunc=cen, want=('count', 'mean'))
    same = lambda a, b: (a == b) | (np.isnan(a) & np.isnan(b))
    tie = ~((lo_run['count'] == ref['count']) & (hi_run['count'] == ref['count']) & same(lo_run['mean'], ref['mean']) & same(hi_run['mean'], ref['mean']))
    assert tie.mean() < 0.5, 'too many tie pixels ' + what
    cnt = r['count'].cpu().numpy()
    assert np.array_equal(np.where(tie, 0, cnt), np.where(tie, 0, ref['count'])), 'count ' + what
    got, want = r['mean'].cpu().numpy(), ref['mean'].astype(np.float32)
    got[tie], want[tie] = 0.0, 0.0
    # a mean near zero of values of scale s carries the float32 sum's absolute error (~1e-7 s): an ulp distance says nothing there
    fin = np.abs(res_ref[np.isfinite(res_ref)])
    scale = float(np.percentile(fin, 99)) if fin.size else 1.0
    small = np.abs(want) < 1e-2 * scale
    assert np.array_equal(np.isnan(got), np.isnan(want)), 'NaN positions ' + what
    assert np.allclose(got[small], want[small], rtol=0, atol=2e-7 * scale, equal_nan=True), 'mean near zero ' + what
    got[small], want[small] = 0.0, 0.0
    assert_ulp(got, want, 1, 'mean ' + what)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--minutes', type=float, default=5.0)
    ap.add_argument('--seed0', type=int, default=100000)
    ap.add_argument('--only', default='', help='comma-separated families: big,stack,image,global,calibrate,frame,resample,arith,fits,chunked,a6,fused')
    a = ap.parse_args()
    t_end = time.time() + 60.0 * a.minutes
    fails, runs = [], {}
    only = set(a.only.split(',')) if a.only else None
    seed = a.seed0
    while time.time() < t_end:
        for name, fn in (('big', big_case), ('stack', lambda s: tf.test_random_stack_configs(ops, apref, s)),
                         ('image', lambda s: tf.test_random_image_kernels(ops, apref, s)), ('global', global_case),
                         ('calibrate', calibrate_case), ('frame', frame_case), ('resample', resample_case), ('arith', arith_case), ('fits', fits_case), ('chunked', chunked_case), ('a6', a6_case), ('fused', fused_case)):
            if only and name not in only:
                continue
            try:
                fn(seed)
            except Exception as ex:                       # noqa: BLE001 - a fuzz driver reports everything
                fails.append((name, seed, str(ex)[:600]))
                print('FAIL', name, seed, str(ex)[:600], flush=True)
                if not isinstance(ex, AssertionError):
                    traceback.print_exc()
            runs[name] = runs.get(name, 0) + 1
        seed += 1
    print('runs', runs, 'failures', len(fails))
    return 1 if fails else 0


if __name__ == '__main__':
    raise SystemExit(main())
