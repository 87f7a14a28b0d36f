#!/usr/bin/env python3
"""Benchmark of the calibrate + sigma-clip-stack hot path on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W

A step = one pass of the hot path over one batch of synthetic frames already resident in HBM:
fused bias/dark/flat calibration + 3-sigma (maxiters 5, median-centred) clipped mean along N of a
64 x 4096 x 4096 float32 slab per GPU (BASELINE.json configs[1], "C2").  With N > 1 GPUs every rank
holds its own 64 frames (weak scaling: the global stack is 64*N frames sharded on the N axis), reduces
them to per-pixel moments and one RCCL all-reduce per row stripe combines them (parallel.stack_nshard).

Prints ONE JSON line (rank 0) with the whole-job Mpixels/s, the HBM roofline of the dominant kernel
(measured live with HIP events on the launch stream) and a CPU baseline (the oracle, OpenMP, bounded
sample) timed beside it.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch
import torch.distributed as dist

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md); 6.3 TB/s is the achievable copy rate


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--frames', type=int, default=64, help='frames per GPU (C2: 64)')
    ap.add_argument('--height', type=int, default=4096)
    ap.add_argument('--width', type=int, default=4096)
    ap.add_argument('--dtype', default='f32', choices=['f32', 'u16'])
    ap.add_argument('--workload', default='c2', choices=['c2', 'c4', 'c5'],
                    help='c2 (default, the BASELINE metric): fused calibrate + clipped mean; c4: uint16 Bayer frames, per-channel '
                         'flat + fused calibrate + median stack (use --height 6248 --width 4176); c5: bad-pixel mask + per-frame '
                         'affine Lanczos-3 resample + 5-iteration clipped mean (use --frames 16 --height 8192 --width 8192)')
    ap.add_argument('--stripes', type=int, default=8, help='row stripes for collective/compute overlap (N > 1)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--force-collective', action='store_true', help='run the striped all-reduce path even with one rank (testing)')
    ap.add_argument('--cpu-seconds', type=float, default=15.0, help='target CPU time of the cpu_baseline sample')
    return ap.parse_args()


def cpu_baseline(frames, masters, nflat, e, seconds):
    """Times the oracle's fused calibrate + clipped stack (oracle/apref.c, OpenMP) on a row sample."""
    import numpy as np
    from oracle import apref
    N, H, W = frames.shape
    threads = apref.num_threads()

    def run(rows):
        sl = slice(0, rows)
        raw = frames[:, sl].cpu().numpy()
        b, d, nf = (masters['bias'][sl].cpu().numpy(), masters['dark'][sl].cpu().numpy(), nflat[sl].cpu().numpy())
        t0 = time.perf_counter()
        apref.calibrate_stack(raw, b, d, nf, e, None, False, sigma=3.0, maxiters=5)
        return time.perf_counter() - t0

    pilot_rows = min(H, 64)
    run(pilot_rows)                                   # warm (library load, page faults, thread pool)
    t = run(pilot_rows)
    rate = N * pilot_rows * W / max(t, 1e-6)          # input pixels / s
    rows = int(max(pilot_rows, min(H, 0.25 * seconds * rate / (N * W))))
    times = []
    t_total = 0.0
    while t_total < seconds and len(times) < 50:
        times.append(run(rows))
        t_total += times[-1]
    times.sort()
    t = times[len(times) // 2]
    model = 'unknown CPU'
    try:
        for ln in open('/proc/cpuinfo'):
            if ln.startswith('model name'):
                model = ln.split(':', 1)[1].strip()
                break
    except OSError:
        pass
    return dict(value=N * rows * W / 1e6 / t, unit='Mpixels/s', cores=threads, kind='port',
                sample='%d frames x %d rows x %d cols f32, median of %d runs (%.1f s of CPU work, OpenMP %d threads on %s, '
                       'oracle/apref.c fused calibrate + clipped stack)' % (N, rows, W, len(times), t_total, threads, model))


def main():
    args = parse()
    # Only the JSON line may appear on stdout: RCCL prints a version banner to the C-level stdout at
    # exit, so everything else (C and Python) is routed to stderr until the final print.
    sys.stdout.flush()
    saved_stdout = os.dup(1)
    os.dup2(2, 1)
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if world > 1 or args.force_collective:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29517')
        os.environ.setdefault('RANK', '0')
        os.environ.setdefault('WORLD_SIZE', '1')
        dist.init_process_group('nccl', device_id=torch.device('cuda', local_rank))
    if args.gpus != world and rank == 0 and world > 1:
        print('warning: --gpus %d but WORLD_SIZE %d' % (args.gpus, world), file=sys.stderr)
    torch.cuda.set_device(local_rank)
    dev = torch.device('cuda', local_rank)

    from astrophotography_amd import ops, synth, parallel

    N, H, W = args.frames, args.height, args.width
    P = H * W
    masters = synth.make_masters(H, W, config_id=2, device=dev)
    nflat, _ = ops.flat_normalize(masters['flat'])
    tdtype = torch.float32 if args.dtype == 'f32' else torch.uint16
    wl = args.workload
    frames = None
    if wl != 'c4':
        frames = synth.make_frames(N, masters, nflat, config_id=2, dtype=tdtype, first_frame=rank * N)
    e = synth.EXP_RATIO
    calib = dict(bias=masters['bias'], dark=masters['dark'], nflat=nflat,
                 exp_ratio=torch.full((N,), e, dtype=torch.float32, device=dev), dark_still_biased=False)
    torch.cuda.synchronize()

    if wl == 'c4':
        if world > 1:
            raise SystemExit('workload c4 is a single-GPU configuration')
        nflat4, _ = ops.bayer_flat_normalize(masters['flat'])
        frames = synth.make_frames(N, masters, nflat4, config_id=4, dtype=torch.uint16, first_frame=0)
        calib = dict(calib, nflat=nflat4)
    if wl == 'c5':
        import numpy as np
        rng = np.random.default_rng(5000 + rank)
        th = np.deg2rad(rng.uniform(-0.2, 0.2, N))
        affines = np.stack([np.cos(th), -np.sin(th), rng.uniform(-3, 3, N), np.sin(th), np.cos(th), rng.uniform(-3, 3, N)], 1)
        cal = ops.calibrate(frames, masters['bias'], masters['dark'], nflat, e)
        del frames
        frames = cal
        st = ops.sigclip_global(masters['dark'], sigma=4.0, maxiters=5)
        badmask, _ = ops.threshold_mask(masters['dark'], thresholds=st[3:5].contiguous())     # lo, hi of the clip, read on the device
        resampled = torch.empty_like(frames)
    torch.cuda.synchronize()

    def step():
        if wl == 'c4':
            return ops.stack_median(frames, calib=calib)
        if wl == 'c5':
            ops.resample_affine(frames, affines, mask=badmask, out=resampled, weight=False)
            if world == 1:
                return ops.stack_sigclip(resampled, sigma=3.0, maxiters=5, outputs=('mean',))['mean']
            return parallel.stack_nshard(resampled, None, sigma=3.0, maxiters=5, n_stripes=args.stripes)
        if world == 1 and not args.force_collective:
            return ops.stack_sigclip(frames, sigma=3.0, maxiters=5, cenfunc='median', stdfunc='std', calib=calib,
                                     outputs=('mean',))['mean']
        return parallel.stack_nshard(frames, calib, sigma=3.0, maxiters=5, n_stripes=args.stripes, force_collective=args.force_collective)

    for _ in range(args.warmup):
        out = step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()

    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    t0 = time.perf_counter()
    for k in range(args.steps):
        evs[k][0].record()
        out = step()
        evs[k][1].record()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    ms_per_step = 1e3 * elapsed / args.steps
    value = world * N * P / 1e6 / (elapsed / args.steps)

    # dominant kernel (stack_sigclip_kernel): device time per launch from HIP events on the launch stream.
    # For world == 1 a step is exactly one launch of it.
    kern_ms = sorted(a.elapsed_time(b) for a, b in evs)
    avg_kernel_ms = sum(kern_ms) / len(kern_ms)
    esize = 4 if args.dtype == 'f32' else 2
    out_planes = 1 if world == 1 else 3
    algo_bytes = esize * N * P + 12 * P + 4 * out_planes * P       # frames + bias/dark/nflat read, outputs written
    kernel_name = 'stack_sigclip_kernel<64,%s,calib>' % ('float' if args.dtype == 'f32' else 'u16')
    metric = 'Mpixels/sec calibrate+sigma-clip-stack'
    workload = 'C2: %dx%dx%d %s per GPU, fused bias/dark/flat + 3-sigma maxiters-5 median-centred clipped mean' % (N, H, W, args.dtype)
    if wl == 'c4':
        algo_bytes = 2 * N * P + 12 * P + 4 * P
        args.dtype = 'u16'
        kernel_name = 'stack_median_u16_kernel<%d,calib>' % N
        metric = 'Mpixels/sec calibrate+median-stack (uint16 Bayer)'
        workload = 'C4: %dx%dx%d u16 RGGB mosaic, per-channel flat normalisation, fused bias/dark/flat + median stack' % (N, H, W)
    if wl == 'c5':
        algo_bytes = (8 * N * P + P) + (4 * N * P + 4 * out_planes * P)     # resample read+write (+mask), stack read + outputs
        kernel_name = 'resample_affine_kernel + stack_sigclip_kernel<%d,float,plain> (step = both launches)' % N
        metric = 'Mpixels/sec mask+affine-resample+sigma-clip-stack'
        workload = 'C5 (per-GPU share): %dx%dx%d f32 calibrated frames, bad-pixel mask, per-frame affine Lanczos-3 resample, 3-sigma maxiters-5 clipped mean' % (N, H, W)
    achieved = algo_bytes / (avg_kernel_ms * 1e-3) / 1e9
    traffic = None
    tfile = os.path.join(ROOT, 'profiles', 'pmc_traffic.json')
    if os.path.exists(tfile) and world == 1 and (N, H, W, args.dtype, wl) == (64, 4096, 4096, 'f32', 'c2'):
        try:
            traffic = json.load(open(tfile)).get('hbm_bytes_per_launch')
        except Exception:
            traffic = None

    # the box's achievable streaming rate beside the nominal peak: device-to-device copy of
    # 1 GiB, read + write bytes over the best of 10 runs
    copy_gbs = None
    if rank == 0:
        src = torch.empty(1 << 28, dtype=torch.float32, device=dev)
        dst = torch.empty_like(src)
        dst.copy_(src)
        best = 1e9
        for _ in range(10):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            dst.copy_(src)
            b.record()
            torch.cuda.synchronize()
            best = min(best, a.elapsed_time(b))
        copy_gbs = 2 * src.numel() * 4 / (best * 1e-3) / 1e9
        del src, dst

    c4file = os.path.join(ROOT, 'profiles', 'r01', 'pmc_c4.json')
    if wl == 'c4' and os.path.exists(c4file) and (N, H, W) == (64, 6248, 4176):
        try:
            d = json.load(open(c4file))
            traffic = d['hbm_read_bytes_fetch_size_x2'] + d['hbm_write_bytes']
        except Exception:
            traffic = None

    line = None
    if rank == 0:
        line = {
            'metric': metric, 'value': value, 'unit': 'Mpixels/s',
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': ms_per_step,
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': args.dtype, 'data': 'synthetic',
            'config': {'workload': workload,
                       'frames_per_gpu': N, 'height': H, 'width': W,
                       'parallelism': 'single GPU' if world == 1 else 'N-shard x%d, %d-stripe all-reduce of sum/sumsq/count' % (world, args.stripes)},
            'roofline': {'bound': 'hbm', 'achieved': achieved, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                         'frac': achieved / HBM_PEAK_GBS, 'traffic': traffic,
                         'kernel': kernel_name,
                         'avg_launch_ms': avg_kernel_ms, 'min_launch_ms': kern_ms[0], 'algorithmic_bytes': algo_bytes,
                         'measured_copy_GBps': copy_gbs, 'frac_of_measured_copy': achieved / copy_gbs if copy_gbs else None},
        }
        if world == 1 and not args.no_cpu_baseline:
            if args.dtype == 'f32' and wl == 'c2':
                line['cpu_baseline'] = cpu_baseline(frames, masters, nflat, e, args.cpu_seconds)
            else:
                line['cpu_baseline'] = None
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()
    sys.stdout.flush()
    try:
        import ctypes
        ctypes.CDLL(None).fflush(None)
    except Exception:
        pass
    os.dup2(saved_stdout, 1)
    if line is not None:
        os.write(1, (json.dumps(line) + '\n').encode())
    # keep late C-level chatter (library destructors) off stdout
    os.dup2(2, 1)


if __name__ == '__main__':
    main()
