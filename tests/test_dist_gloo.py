"""world_size-2 and -4 gloo tests (CPU) of the multi-GPU plumbing in astrophotography_amd.parallel.

The per-rank partial moments and the finalisation are HIP kernels in the product; here CPU stand-ins
built on the oracle are injected so that sharding, striping, the all-reduces (both exchange payloads) and
the assembly of the result run for real under torch.distributed (gloo).  The row-shard test exercises the
partition / gather helpers the same way."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _oracle_local_moments(frames, calib, r0, r1, clip, exchange):
    """Stand-in for ops.stack_sigclip(outputs=('moments' | 'moments_f64p')): the same planes, computed by the oracle."""
    from oracle import apref
    sub = frames[:, r0:r1].numpy()
    if calib is not None:
        nflat = calib['nflat'][r0:r1].numpy() if calib.get('nflat') is not None else None
        sub = apref.calibrate(sub, calib['bias'][r0:r1].numpy(), calib['dark'][r0:r1].numpy(), nflat, calib['exp_ratio'],
                              calib.get('pedestal'), calib.get('dark_still_biased', False))
    r = apref.stack_sigclip(sub, sigma=clip['sigma'], maxiters=clip['maxiters'], cenfunc=clip['cenfunc'],
                            stdfunc=clip['stdfunc'], want=('keep', 'count'))
    kept = np.where(r['keep'], sub.astype(np.float64), 0.0)
    if exchange == 'f32':
        mom = torch.from_numpy(np.stack([kept.sum(0), r['count'].astype(np.float64)]).astype(np.float32))
        return dict(sum=mom[0], count=mom[1], prefix=mom)
    if exchange == 'f64i':                                     # the layout of 'rs': float64 sum / sumsq planes + an int32 count plane
        return dict(sum=torch.from_numpy(kept.sum(0)), sumsq=torch.from_numpy((kept * kept).sum(0)),
                    count=torch.from_numpy(r['count'].astype(np.int32)))
    # the packed float64 layout: planes sum, count, sumsq in ONE buffer; a mean-only exchange sends the [2] prefix
    buf = torch.from_numpy(np.stack([kept.sum(0), r['count'].astype(np.float64), (kept * kept).sum(0)]))
    return dict(sum=buf[0], count=buf[1], sumsq=buf[2], buffer=buf, prefix=buf[:2])


def _cpu_finalize(m, out_mean, out_std, exchange):
    if exchange == 'f32':
        out_mean.copy_(m['sum'] / m['count'])
        return
    n = m['count'].double()
    mean = m['sum'] / n
    out_mean.copy_(mean.float())
    if out_std is not None:
        out_std.copy_((m['sumsq'] / n - mean * mean).clamp_min(0).sqrt().float())


def _data(n_total, shape):
    rng = np.random.default_rng(42)                                  # same data on every rank
    cube = rng.normal(50000, 30, (n_total,) + shape).astype(np.float32)   # CCD-range values: float32 sums of squares would cancel
    hits = rng.random(cube.shape) < 0.03
    cube[hits] += 3000
    bias = rng.normal(100, 2, shape).astype(np.float32)
    dark = rng.normal(10, 1, shape).astype(np.float32)
    return cube, bias, dark


def _worker(rank, world, port, n_total, shape, out_dir):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from astrophotography_amd import parallel
    cube, bias, dark = _data(n_total, shape)
    lo, hi = parallel.shard_frames(n_total, world, rank)
    calib = dict(bias=torch.from_numpy(bias), dark=torch.from_numpy(dark), nflat=None, exp_ratio=0.4)
    mine = torch.from_numpy(cube[lo:hi])
    kw = dict(sigma=3.0, maxiters=5, local_moments=_oracle_local_moments, finalize=_cpu_finalize)
    calls = []
    real_all_reduce = dist.all_reduce

    def counting_all_reduce(t, *a, **k):
        calls.append((tuple(t.shape), t.dtype))
        return real_all_reduce(t, *a, **k)
    parallel.dist.all_reduce = counting_all_reduce
    (mean, std), parts = parallel.stack_nshard(mine, calib, n_stripes=3, exchange='f64', want_std=True, return_moments=True, **kw)
    # ONE all-reduce per stripe, carrying all three float64 planes
    assert len(calls) == 3 and all(sh[0] == 3 and dt == torch.float64 for sh, dt in calls), calls
    del calls[:]
    mean_b = parallel.stack_nshard(mine, calib, n_stripes=4, exchange='f64', **kw)          # mean-only exchange: same mean
    assert torch.equal(mean, mean_b)
    assert len(calls) == 4 and all(sh[0] == 2 and dt == torch.float64 for sh, dt in calls), calls      # mean only: the [2] prefix
    del calls[:]
    mean32 = parallel.stack_nshard(mine, calib, n_stripes=2, exchange='f32', **kw)
    assert len(calls) == 2 and all(sh[0] == 2 and dt == torch.float32 for sh, dt in calls), calls
    parallel.dist.all_reduce = real_all_reduce
    assert parallel.exchange_bytes_per_pixel('f64') == 16 and parallel.exchange_bytes_per_pixel('f64', True) == 24
    assert parallel.default_stripes(4096, 4096) == 4 and parallel.default_stripes(64, 64) == 1
    with pytest.raises(ValueError):
        parallel.stack_nshard(mine, calib, exchange='f32', want_std=True, **kw)
    cnt = torch.cat([p['count'] for p in parts], 0).to(torch.int64)
    np.savez(os.path.join(out_dir, f'r{rank}.npz'), mean=mean.numpy(), std=std.numpy(), mean32=mean32.numpy(), count=cnt.numpy())

    # row-shard: every rank reduces all frames of its own rows (stand-in for the device call), then the image is gathered
    H = shape[0]
    r0, r1 = parallel.row_block(H, world, rank)
    from oracle import apref
    cal = apref.calibrate(cube[:, r0:r1], bias[r0:r1], dark[r0:r1], None, 0.4)
    block = torch.from_numpy(apref.stack_sigclip(cal, sigma=3.0, maxiters=5, want=('mean',))['mean'].astype(np.float32))
    full = parallel.gather_rows(block, H)
    np.save(os.path.join(out_dir, f'rows{rank}.npy'), full.numpy())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(600)
@pytest.mark.parametrize('world,n_total', [(2, 24), (4, 50)])         # 50 frames on 4 ranks: ragged shards 13 + 13 + 12 + 12
def test_nshard_and_rowshard_gloo(tmp_path, world, n_total):
    from oracle import apref
    from astrophotography_amd import parallel
    shape = (11, 16)                                                  # 11 rows: ragged row blocks and stripes
    port = _free_port()
    mp.spawn(_worker, args=(world, port, n_total, shape, str(tmp_path)), nprocs=world, join=True)
    res = [np.load(tmp_path / f'r{r}.npz') for r in range(world)]
    for k in ('mean', 'std', 'mean32', 'count'):
        for r in range(1, world):
            assert np.array_equal(res[0][k], res[r][k]), k            # every rank holds the result
    # expected: hierarchical clipping = sigma-clip each rank's frames, add the float64 moments (SURVEY 8(e) option ii)
    cube, bias, dark = _data(n_total, shape)
    cal = apref.calibrate(cube, bias, dark, None, 0.4)
    tot = np.zeros(shape, np.float64)
    tot2 = np.zeros(shape, np.float64)
    cnt = np.zeros(shape, np.int64)
    for r in range(world):
        lo, hi = parallel.shard_frames(n_total, world, r)
        rr = apref.stack_sigclip(cal[lo:hi], sigma=3.0, maxiters=5, want=('keep', 'count'))
        kept = np.where(rr['keep'], cal[lo:hi].astype(np.float64), 0)
        tot += kept.sum(0)
        tot2 += (kept * kept).sum(0)
        cnt += rr['count']
    assert np.array_equal(res[0]['count'], cnt)
    mean64 = tot / cnt
    # float64 exchange: the float64 combine rounded ONCE - bit-equal to the float32 rounding of the float64 mean
    assert np.array_equal(res[0]['mean'], mean64.astype(np.float32))
    std64 = np.sqrt(np.maximum(tot2 / cnt - mean64 * mean64, 0))
    np.testing.assert_allclose(res[0]['std'], std64, rtol=1e-6)
    if n_total // world >= 12:                                        # (a 3-sigma clip cannot reject one outlier among 6 or 7
        assert 20 < np.median(res[0]['std']) < 40                     # values) sigma 30 on a 50000 ADU level survives the combine
    # float32 exchange: per-rank rounding of the sums - close, not exact
    np.testing.assert_allclose(res[0]['mean32'], mean64, rtol=3e-7)
    assert cnt.min() >= n_total - 8 and cnt.max() == n_total and (cnt < n_total).any()   # outliers were clipped
    # row-shard + gather = the exact full-stack result on every rank
    full_ref = apref.stack_sigclip(cal, sigma=3.0, maxiters=5, want=('mean',))['mean'].astype(np.float32)
    for r in range(world):
        assert np.array_equal(np.load(tmp_path / f'rows{r}.npy'), full_ref)


def _rs_worker(rank, world, port, n_total, shape, out_dir):
    """exchange='rs' against the all-reduce form on the same data: collective kinds, shapes, dtypes and bytes."""
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from astrophotography_amd import parallel
    cube, bias, dark = _data(n_total, shape)
    lo, hi = parallel.shard_frames(n_total, world, rank)
    calib = dict(bias=torch.from_numpy(bias), dark=torch.from_numpy(dark), nflat=None, exp_ratio=0.4)
    mine = torch.from_numpy(cube[lo:hi])
    kw = dict(sigma=3.0, maxiters=5, local_moments=_oracle_local_moments, finalize=_cpu_finalize)
    calls = []
    real = dict(ar=dist.all_reduce, rs=dist.reduce_scatter_tensor, ag=dist.all_gather_into_tensor)

    def ar(t, *a, **k):
        calls.append(('all_reduce', tuple(t.shape), t.dtype, t.numel() * t.element_size()))
        return real['ar'](t, *a, **k)

    def rs(out, inp, *a, **k):
        calls.append(('reduce_scatter', tuple(inp.shape), inp.dtype, inp.numel() * inp.element_size()))
        return real['rs'](out, inp, *a, **k)

    def ag(out, inp, *a, **k):
        calls.append(('all_gather', tuple(out.shape), out.dtype, out.numel() * out.element_size()))
        return real['ag'](out, inp, *a, **k)
    parallel.dist.all_reduce, parallel.dist.reduce_scatter_tensor, parallel.dist.all_gather_into_tensor = ar, rs, ag
    H, W = shape
    n_stripes = 2
    h = H // n_stripes
    # the default exchange IS 'rs'
    mean_rs, parts = parallel.stack_nshard(mine, calib, n_stripes=n_stripes, return_moments=True, **kw)
    # per stripe: a reduce-scatter of the float64 sum plane [h, W] and one of the INT32 count plane, one all-gather of the float32 mean rows
    exp = []
    for _ in range(n_stripes):
        exp += [('reduce_scatter', (h, W), torch.float64, 8 * h * W), ('reduce_scatter', (h, W), torch.int32, 4 * h * W),
                ('all_gather', (h, W), torch.float32, 4 * h * W)]
    assert calls == exp, calls
    assert all(p['sum'].shape == (h // world, W) and p['count'].dtype == torch.int32 and
               p['rows'] == (rank * (h // world), (rank + 1) * (h // world)) for p in parts)
    del calls[:]
    # the count as a float16 plane (exact up to 2048 frames in the whole job): 2 bytes per pixel, the same result bit for bit
    mean_h = parallel.stack_nshard(mine, calib, n_stripes=n_stripes, count_dtype=torch.float16, **kw)
    assert [c[2] for c in calls if c[0] == 'reduce_scatter'] == [torch.float64, torch.float16] * n_stripes, calls
    assert sum(c[3] for c in calls) == n_stripes * (8 + 2 + 4) * h * W
    assert torch.equal(mean_h, mean_rs)
    del calls[:]
    # gather=False: the result stays row-distributed - two reduce-scatters per stripe and NO all-gather (10 bytes per pixel with
    # the float16 count); the rank's rows (parallel.own_rows) equal the gathered form's, bit for bit
    mean_d = parallel.stack_nshard(mine, calib, n_stripes=n_stripes, count_dtype=torch.float16, gather=False, **kw)
    assert [c[0] for c in calls] == ['reduce_scatter'] * (2 * n_stripes), calls
    assert sum(c[3] for c in calls) == n_stripes * (8 + 2) * h * W
    rows = parallel.own_rows(H, n_stripes, world, rank)
    assert rows == [(k * h + rank * (h // world), k * h + (rank + 1) * (h // world)) for k in range(n_stripes)]
    for a0, b0 in rows:
        assert torch.equal(mean_d[a0:b0], mean_rs[a0:b0])
    assert parallel.exchange_bytes_on_wire('rs', world, H * W, count_bytes=2, gather=False) == int((world - 1) / world * 10 * H * W)
    try:
        parallel.stack_nshard(mine, calib, n_stripes=3, gather=False, **kw)          # 6, 5, 5 rows: not divisible
        assert world == 1
    except ValueError:
        pass
    del calls[:]
    (mean_s, std_s), _ = parallel.stack_nshard(mine, calib, n_stripes=n_stripes, exchange='rs', want_std=True, return_moments=True, **kw)
    assert [c[0] for c in calls] == (['reduce_scatter'] * 3 + ['all_gather'] * 2) * n_stripes, calls
    del calls[:]
    (mean_ar, std_ar) = parallel.stack_nshard(mine, calib, n_stripes=n_stripes, exchange='f64', want_std=True, **kw)
    assert [c[0] for c in calls] == ['all_reduce'] * n_stripes and all(c[1][0] == 3 for c in calls), calls
    del calls[:]
    # a stripe whose rows do not divide by the world size takes the all-reduce form (here: 3 stripes of 16 rows -> 6, 5, 5)
    mean_rag = parallel.stack_nshard(mine, calib, n_stripes=3, exchange='rs', **kw)
    kinds = [c[0] for c in calls]
    assert 'all_reduce' in kinds or world in (1,), kinds
    parallel.dist.all_reduce, parallel.dist.reduce_scatter_tensor, parallel.dist.all_gather_into_tensor = real['ar'], real['rs'], real['ag']
    # bytes a rank sends per step (ring collectives)
    f = (world - 1) / world
    assert parallel.exchange_bytes_on_wire('rs', world, H * W) == int(f * 16 * H * W)                   # 14 bytes per pixel at 8 ranks
    assert parallel.exchange_bytes_on_wire('rs', world, H * W, count_bytes=2) == int(f * 14 * H * W)    # 12.25 with the float16 count
    assert parallel.exchange_bytes_on_wire('rs', world, H * W, want_std=True) == int(f * 28 * H * W)
    assert parallel.default_stripes(4096, 4096, 'rs') == 4 and parallel.default_stripes(8192, 8192, 'rs') == 8
    assert parallel.exchange_bytes_on_wire('f64', world, H * W) == int(2 * f * 16 * H * W)
    assert parallel.exchange_bytes_on_wire('f32', world, H * W) == int(2 * f * 8 * H * W)
    np.savez(os.path.join(out_dir, f'rs{rank}.npz'), mean_rs=mean_rs.numpy(), mean_s=mean_s.numpy(), std_s=std_s.numpy(),
             mean_ar=mean_ar.numpy(), std_ar=std_ar.numpy(), mean_rag=mean_rag.numpy())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(600)
@pytest.mark.parametrize('world,n_total', [(2, 24), (4, 50), (8, 96)])
def test_nshard_reduce_scatter_exchange_gloo(tmp_path, world, n_total):
    """exchange='rs' (reduce-scatter of the float64 planes by rows + all-gather of the float32 result rows): collective
    count / shapes / bytes for world 2, 4 and 8, every rank holds the whole image, and the result EQUALS the all-reduce
    form's (the same float64 sums, added by a different collective; on these data bit for bit)."""
    shape = (16, 16)                                                  # 2 stripes of 8 rows: divisible by 2, 4 and 8
    port = _free_port()
    mp.spawn(_rs_worker, args=(world, port, n_total, shape, str(tmp_path)), nprocs=world, join=True)
    res = [np.load(tmp_path / f'rs{r}.npz') for r in range(world)]
    for k in res[0].files:
        for r in range(1, world):
            assert np.array_equal(res[0][k], res[r][k]), k
    a = res[0]
    # float64 sums of at most 8 addends per pixel: the order of the additions can move the float64 sum by an ulp, the
    # float32 mean by at most one ulp and almost never
    for x, y in ((a['mean_rs'], a['mean_ar']), (a['mean_s'], a['mean_ar']), (a['mean_rag'], a['mean_ar'])):
        ulp = np.abs(x.view(np.int32).astype(np.int64) - y.view(np.int32).astype(np.int64))
        assert ulp.max() <= 1 and (ulp == 0).mean() > 0.99
    np.testing.assert_allclose(a['std_s'], a['std_ar'], rtol=1e-6)
