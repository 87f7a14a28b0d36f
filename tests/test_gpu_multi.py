"""Two-GPU tests (skipped on a one-GPU box): the N-shard path with real RCCL against "oracle per shard, float64 moments
added" (SURVEY 8(e) option ii), the row-shard path against the full-stack oracle, and bench.py's own launcher."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _ngpu():
    try:
        return torch.cuda.device_count()
    except Exception:
        return 0


needs2 = pytest.mark.skipif(_ngpu() < 2, reason='needs two GPUs')


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _rank_main(rank, world, port, out_dir):
    """One rank of the 2-GPU job (fresh process: started with the spawn method before any GPU call)."""
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from astrophotography_amd import ops, parallel
    torch.cuda.set_device(rank)
    dist.init_process_group('nccl', init_method=f'tcp://127.0.0.1:{port}', rank=rank, world_size=world,
                            device_id=torch.device('cuda', rank))
    d = np.load(os.path.join(out_dir, 'data.npz'))
    cube, bias, dark, nflat = d['cube'], d['bias'], d['dark'], d['nflat']
    N, H, W = cube.shape
    dev = torch.device('cuda', rank)
    lo, hi = parallel.shard_frames(N, world, rank)
    calib = dict(bias=torch.from_numpy(bias).to(dev), dark=torch.from_numpy(dark).to(dev), nflat=torch.from_numpy(nflat).to(dev),
                 exp_ratio=0.4)
    mine = torch.from_numpy(cube[lo:hi]).to(dev)
    mean64, std64 = parallel.stack_nshard(mine, calib, n_stripes=3, exchange='f64', want_std=True)
    # the reduce-scatter form (the default): 37 rows in 1 stripe do not divide by 2 -> all-reduce fallback; a 36-row cut does
    mean_rs, std_rs = parallel.stack_nshard(mine[:, :36], parallel._slice_calib(calib, 0, 36), n_stripes=2, want_std=True)
    mean32 = parallel.stack_nshard(mine, calib, n_stripes=4, exchange='f32')
    r0, r1 = parallel.row_block(H, world, rank)
    rows = torch.from_numpy(cube[:, r0:r1]).to(dev)
    blk = parallel.stack_rowshard(rows, parallel._slice_calib(calib, r0, r1))['mean']
    full = parallel.gather_rows(blk, H)
    torch.cuda.synchronize()
    np.savez(os.path.join(out_dir, f'r{rank}.npz'), mean64=mean64.cpu().numpy(), std64=std64.cpu().numpy(),
             mean32=mean32.cpu().numpy(), rows=full.cpu().numpy(), mean_rs=mean_rs.cpu().numpy(), std_rs=std_rs.cpu().numpy())
    dist.barrier()
    dist.destroy_process_group()


@needs2
@pytest.mark.timeout(600)
def test_nshard_and_rowshard_two_gpus_rccl(tmp_path):
    import torch.multiprocessing as mp
    from oracle import apref
    from astrophotography_amd import parallel
    from tests.util import assert_ulp
    rng = np.random.default_rng(7)
    N, H, W = 24, 37, 256
    cube = rng.normal(50000, 30, (N, H, W)).astype(np.float32)
    cube[rng.random(cube.shape) < 0.02] += 4000
    bias = rng.normal(100, 2, (H, W)).astype(np.float32)
    dark = rng.normal(10, 1, (H, W)).astype(np.float32)
    nflat = rng.normal(1.0, 0.02, (H, W)).astype(np.float32)
    np.savez(tmp_path / 'data.npz', cube=cube, bias=bias, dark=dark, nflat=nflat)
    world = 2
    mp.get_context('spawn')
    mp.spawn(_rank_main, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    res = [np.load(tmp_path / f'r{r}.npz') for r in range(world)]
    for k in res[0].files:
        assert np.array_equal(res[0][k], res[1][k], equal_nan=True), k
    cal = apref.calibrate(cube, bias, dark, nflat, 0.4)
    tot = np.zeros((H, W))
    tot2 = np.zeros((H, W))
    cnt = np.zeros((H, W), np.int64)
    for r in range(world):
        lo, hi = parallel.shard_frames(N, world, r)
        rr = apref.stack_sigclip(cal[lo:hi], sigma=3.0, maxiters=5, want=('keep', 'count'))
        kept = np.where(rr['keep'], cal[lo:hi].astype(np.float64), 0)
        tot += kept.sum(0)
        tot2 += (kept * kept).sum(0)
        cnt += rr['count']
    mean_ref = tot / cnt
    assert_ulp(res[0]['mean64'], mean_ref.astype(np.float32), 1, 'float64 exchange vs the float64 combine')
    np.testing.assert_allclose(res[0]['mean32'], mean_ref, rtol=3e-7)
    np.testing.assert_allclose(res[0]['std64'], np.sqrt(np.maximum(tot2 / cnt - mean_ref ** 2, 0)), rtol=1e-5)
    assert_ulp(res[0]['mean_rs'], mean_ref[:36].astype(np.float32), 1, 'reduce-scatter exchange vs the float64 combine')
    np.testing.assert_allclose(res[0]['std_rs'], np.sqrt(np.maximum(tot2 / cnt - mean_ref ** 2, 0))[:36], rtol=1e-5)
    full_ref = apref.stack_sigclip(cal, sigma=3.0, maxiters=5, want=('mean',))['mean'].astype(np.float32)
    assert_ulp(res[0]['rows'], full_ref, 1, 'row-shard + gather vs the full-stack oracle')


@needs2
@pytest.mark.timeout(900)
def test_bench_launcher_two_gpus():
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_PORT', 'MASTER_ADDR')}
    base = [sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '3', '--warmup', '1', '--height', '256',
            '--width', '512', '--frames', '16']
    for extra in ([], ['--exchange', 'f32'], ['--exchange', 'f64'], ['--scaling', 'strong', '--total-frames', '32'], ['--parallelism', 'rowshard']):
        r = subprocess.run(base + extra, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=800)
        assert r.returncode == 0, r.stderr[-2000:]
        lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
        assert len(lines) == 1, r.stdout
        d = json.loads(lines[0])
        assert d['n_gpus'] == 2 and d['rccl_world_size'] == 2 and len(d['per_rank_ms']) == 2 and d['value'] > 0
