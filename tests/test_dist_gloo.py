"""world_size-2 gloo test (CPU) of the N-shard collective plumbing in astrophotography_amd.parallel.

The per-rank partial moments and the finalisation are HIP kernels in the product; here CPU stand-ins
built on the oracle are injected so that sharding, striping, the all-reduce and the assembly of the
result run for real under torch.distributed (gloo)."""
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


def _oracle_local_moments(frames, calib, r0, r1, clip):
    from oracle import apref
    sub = frames[:, r0:r1].numpy()
    if calib is not None:
        nflat = calib['nflat'][r0:r1].numpy() if calib.get('nflat') is not None else None
        sub = apref.calibrate(sub, calib['bias'][r0:r1].numpy(), calib['dark'][r0:r1].numpy(), nflat, calib['exp_ratio'],
                              calib.get('pedestal'), calib.get('dark_still_biased', False))
    r = apref.stack_sigclip(sub, sigma=clip['sigma'], maxiters=clip['maxiters'], cenfunc=clip['cenfunc'],
                            stdfunc=clip['stdfunc'], want=('keep', 'count'))
    kept = np.where(r['keep'], sub.astype(np.float64), 0.0)
    mom = np.stack([kept.sum(0), r['count'].astype(np.float64), (kept * kept).sum(0)]).astype(np.float32)    # sum, count, sumsq
    return torch.from_numpy(mom)


def _cpu_finalize(moments, out_mean):
    out_mean.copy_(moments[0] / moments[1])


def _worker(rank, world, port, n_total, shape, out_dir):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from astrophotography_amd import parallel
    rng = np.random.default_rng(42)                                  # same data on every rank
    cube = rng.normal(500, 20, (n_total,) + shape).astype(np.float32)
    hits = rng.random(cube.shape) < 0.03
    cube[hits] += 3000
    bias = rng.normal(100, 2, shape).astype(np.float32)
    dark = rng.normal(10, 1, shape).astype(np.float32)
    lo, hi = parallel.shard_frames(n_total, world, rank)
    calib = dict(bias=torch.from_numpy(bias), dark=torch.from_numpy(dark), nflat=None, exp_ratio=0.4)
    mean, mom = parallel.stack_nshard(torch.from_numpy(cube[lo:hi]), calib, sigma=3.0, maxiters=5, n_stripes=3,
                                      local_moments=_oracle_local_moments, finalize=_cpu_finalize, return_moments=True)
    # the mean-only exchange all-reduces just the (sum, count) prefix of the moments: same mean
    mean2 = parallel.stack_nshard(torch.from_numpy(cube[lo:hi]), calib, sigma=3.0, maxiters=5, n_stripes=4,
                                  local_moments=_oracle_local_moments, finalize=_cpu_finalize)
    assert torch.equal(mean, mean2)
    np.save(os.path.join(out_dir, f'mean{rank}.npy'), mean.numpy())
    np.save(os.path.join(out_dir, f'mom{rank}.npy'), mom.numpy())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_nshard_allreduce_world2(tmp_path):
    from oracle import apref
    world, n_total, shape = 2, 24, (10, 16)
    port = _free_port()
    mp.spawn(_worker, args=(world, port, n_total, shape, str(tmp_path)), nprocs=world, join=True)
    means = [np.load(tmp_path / f'mean{r}.npy') for r in range(world)]
    moms = [np.load(tmp_path / f'mom{r}.npy') for r in range(world)]
    assert np.array_equal(means[0], means[1]) and np.array_equal(moms[0], moms[1])     # every rank holds the result
    # expected: hierarchical clipping = sigma-clip each rank's frames, add the moments (SURVEY 8(e) option ii)
    rng = np.random.default_rng(42)
    cube = rng.normal(500, 20, (n_total,) + shape).astype(np.float32)
    hits = rng.random(cube.shape) < 0.03
    cube[hits] += 3000
    bias = rng.normal(100, 2, shape).astype(np.float32)
    dark = rng.normal(10, 1, shape).astype(np.float32)
    cal = apref.calibrate(cube, bias, dark, None, 0.4)
    tot = np.zeros(shape, np.float64)
    cnt = np.zeros(shape, np.float64)
    for r in range(world):
        lo, hi = (0, 12) if r == 0 else (12, 24)
        rr = apref.stack_sigclip(cal[lo:hi], sigma=3.0, maxiters=5, want=('keep', 'count'))
        tot += np.where(rr['keep'], cal[lo:hi].astype(np.float64), 0).sum(0)
        cnt += rr['count']
    assert np.array_equal(moms[0][1], cnt.astype(np.float32))                           # planes: sum, count, sumsq
    np.testing.assert_allclose(means[0], tot / cnt, rtol=3e-7)
    assert cnt.min() >= 16 and cnt.max() == 24 and (cnt < 24).any()                     # outliers were clipped
