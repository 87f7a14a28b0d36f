"""Development aid: re-run seeds of the `fused` fuzz family and compare fused / two-step / oracle pixel by pixel."""
import sys
import numpy as np
import torch
sys.path.insert(0, '.')
from astrophotography_amd import ops
from oracle import apref
import tools.fuzz_long as fz


def case(seed):
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
    if rng.integers(0, 4) == 0:
        ty, tx = (h + 15) // 16, (w + 63) // 64
        A = np.repeat(np.repeat(A[:, None, None, :], ty, 1), tx, 2).copy()
        A[..., 2] += rng.uniform(-0.2, 0.2, A.shape[:-1])
        A[..., 5] += rng.uniform(-0.2, 0.2, A.shape[:-1])
    mask = (rng.random((H, W)) < float(rng.choice([1e-3, 0.02, 0.2]))).astype(np.uint8) if rng.integers(0, 2) else None
    fs = rng.uniform(0.1, 3.0, N).astype(np.float32) if rng.integers(0, 2) else None
    nph = int(rng.choice([64, 1024, 4096]))
    cf = bool(rng.integers(0, 2))
    sigma = float(rng.choice([1.5, 2.0, 3.0, 5.0]))
    maxiters = [1, 2, 5, None][rng.integers(0, 4)]
    cen = str(rng.choice(['median', 'mean']))
    exact = bool(rng.integers(0, 3) == 0)
    return dict(frames=frames, A=A, out_shape=out_shape, mask=mask, fs=fs, nph=nph, cf=cf, sigma=sigma, maxiters=maxiters, cen=cen, exact=exact, N=N)


for seed in [int(a) for a in sys.argv[1:]]:
    c = case(seed)
    fr = torch.from_numpy(c['frames']).cuda()
    mk = None if c['mask'] is None else torch.from_numpy(c['mask']).cuda()
    with np.errstate(all='ignore'):
        res_ref, _ = apref.resample_affine(c['frames'], c['A'], fscale=c['fs'], mask=c['mask'], out_shape=c['out_shape'], n_phases=c['nph'], conserve_flux=c['cf'])
        ref = apref.stack_sigclip(res_ref, sigma=c['sigma'], maxiters=c['maxiters'], cenfunc=c['cen'])
    f = ops.resample_stack_sigclip(fr, c['A'], fscale=c['fs'], mask=mk, out_shape=c['out_shape'], n_phases=c['nph'], conserve_flux=c['cf'],
                                   sigma=c['sigma'], maxiters=c['maxiters'], cenfunc=c['cen'], outputs=('mean', 'count'), exact=c['exact'])
    r2, _ = ops.resample_affine(fr, c['A'], fscale=c['fs'], mask=mk, out_shape=c['out_shape'], n_phases=c['nph'], conserve_flux=c['cf'], weight=False)
    t = ops.stack_sigclip(r2, sigma=c['sigma'], maxiters=c['maxiters'], cenfunc=c['cen'], outputs=('mean', 'count'), exact=c['exact'])
    fc, tc, rc = f['count'].cpu().numpy(), t['count'].cpu().numpy(), ref['count']
    print('seed', seed, 'N', c['N'], 'sigma', c['sigma'], 'maxiters', c['maxiters'], c['cen'], 'exact', c['exact'], 'shape', rc.shape,
          '| fused != oracle:', int((fc != rc).sum()), ' two-step != oracle:', int((tc != rc).sum()), ' fused != two-step:', int((fc != tc).sum()),
          ' resampled equal:', bool(np.array_equal(r2.cpu().numpy(), res_ref, equal_nan=True)))
    fm, tm, rm = f['mean'].cpu().numpy(), t['mean'].cpu().numpy(), ref['mean'].astype(np.float32)
    def ulp(a, b):
        ia, ib = a.view(np.int32).astype(np.int64), b.view(np.int32).astype(np.int64)
        ia = np.where(ia < 0, -(ia & 0x7fffffff), ia); ib = np.where(ib < 0, -(ib & 0x7fffffff), ib)
        d = np.abs(ia - ib); d[np.isnan(a) | np.isnan(b)] = 0
        return d
    du, dt, dft = ulp(fm, rm), ulp(tm, rm), ulp(fm, tm)
    print('   mean ulp: fused vs oracle max', du.max(), '(>1:', int((du > 1).sum()), ') two-step vs oracle max', dt.max(), '(>1:', int((dt > 1).sum()), ') fused vs two-step max', dft.max())
    for y, x in np.argwhere(du > 1)[:3]:
        col = res_ref[:, y, x]
        print('   pixel', y, x, 'fused', repr(fm[y, x]), 'two-step', repr(tm[y, x]), 'oracle f64', repr(ref['mean'][y, x]), 'count', rc[y, x], 'column', np.sort(col[np.isfinite(col)]).tolist())
    bad = np.argwhere(fc != rc)[:3]
    for y, x in bad:
        col = res_ref[:, y, x]
        print('   pixel', y, x, 'fused', fc[y, x], 'two-step', tc[y, x], 'oracle', rc[y, x], 'column', np.sort(col[np.isfinite(col)]).tolist())
