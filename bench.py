#!/usr/bin/env python3
"""Benchmark of the calibrate + sigma-clip-stack hot path on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W [--scaling weak|strong] [--parallelism nshard|rowshard]

A step = one pass of the hot path over one batch of synthetic frames already resident in HBM: fused
bias/dark/flat calibration + 3-sigma (maxiters 5, median-centred) clipped mean along N.

  N = 1 (default)        BASELINE.json configs[1] "C2": 64 x 4096 x 4096 float32, one launch of the fused kernel.
  N > 1, --scaling weak  (default) every rank holds its own 64 x 4096 x 4096 frames: the global stack is 64*N frames
                         sharded on the N axis (parallel.stack_nshard: per-rank partial moments, one RCCL all-reduce
                         per row stripe overlapped with the reduction of the next stripes).
  --scaling strong       BASELINE.json configs[2] "C3": --total-frames (256) x 4096 x 4096 in total, 256/N per rank, with the
                         SAME semantics at every N: the job is --hier-shards (8) shards of 32 frames, each clipped against
                         its own statistics, float64 moments added - a rank holding several shards reduces them chunk by
                         chunk on the device.  At N = 1 the exact 256-frame kernel is timed beside it ("exact_ms").
  N > 1                  the line also carries the row-shard time of the same job ("rowshard") and the time the stripes'
                         all-reduces take on the communication stream ("exchange_ms").
  --parallelism rowshard   the exact partition: every rank reduces ALL frames of its own row block, no data-path
                         collective (weak: 4096 rows per rank, strong: 4096/N rows per rank).

Launch: `python bench.py --gpus N` starts N fresh child processes itself (one per GPU, before anything in the
parent touches the GPU) unless it already runs under a launcher (torchrun sets WORLD_SIZE); the ranks rendezvous over
RCCL (torch.distributed backend "nccl") on 127.0.0.1.  The run fails (non-zero exit) if the process group's world size
differs from --gpus.

Prints ONE JSON line (rank 0) with the whole-job Mpixels/s, the HBM roofline of the dominant kernel (measured live
with HIP events on the launch stream) and the CPU baselines timed beside it (the C/OpenMP oracle and the NumPy path).
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md); 6.3 TB/s is the achievable copy rate


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--frames', type=int, default=None, help='frames per GPU (weak scaling); default: the BASELINE configuration of '
                                                            'the workload (C2: 64, C4: 64, C5: 16 = the per-GPU share of 128)')
    ap.add_argument('--total-frames', type=int, default=256, help='frames of the whole job with --scaling strong (C3: 256)')
    ap.add_argument('--height', type=int, default=None, help='default: BASELINE (C2: 4096, C4: 6248, C5: 8192)')
    ap.add_argument('--width', type=int, default=None, help='default: BASELINE (C2: 4096, C4: 4176, C5: 8192)')
    ap.add_argument('--dtype', default='f32', choices=['f32', 'u16'])
    ap.add_argument('--scaling', default='weak', choices=['weak', 'strong'])
    ap.add_argument('--parallelism', default='nshard', choices=['nshard', 'rowshard'])
    ap.add_argument('--exchange', default='rs', choices=['rs', 'f64', 'f32'],
                    help='N-shard exchange: rs (default) = packed float64 sum + count planes reduce-scattered by rows, every rank '
                         'finalises its rows, float32 mean rows all-gathered (17.5 B/pixel on the wire at 8 ranks); f64 = ONE '
                         'all-reduce per stripe of the same planes (16 B/pixel payload, 28 on the wire; the north star\'s literal '
                         'form); f32 = one all-reduce of float32 sum + count (8 B/pixel payload)')
    ap.add_argument('--rs-count', default='auto', choices=['auto', 'f16', 'i32'],
                    help="exchange rs: the count plane's type on the wire - f16 (2 bytes, exact up to 2048 frames in the whole job), "
                         'i32, or auto = f16 when the job has at most 2048 frames')
    ap.add_argument('--workload', default='c2', choices=['c2', 'c4', 'c5'],
                    help='c2 (default, the BASELINE metric): fused calibrate + clipped mean, 64 x 4096 x 4096 f32; c4: uint16 Bayer '
                         'frames, per-channel flat + fused calibrate + median stack, 64 x 6248 x 4176; c5: bad-pixel mask + per-frame '
                         'affine Lanczos-3 resample + 5-iteration clipped mean, the per-GPU share 16 x 8192 x 8192 of 128 frames '
                         '(a bare --workload runs BASELINE\'s dimensions; --frames / --height / --width override them)')
    ap.add_argument('--c5-dithered', action='store_true', help='workload c5: the synthetic frames are a DITHERED sequence - frame f images the '
                    'scene warped by the inverse of its registration transform, so that the resample aligns the stars (default: every frame '
                    'images the scene at the same place and the transforms MISalign it: star pixels then fail the clip wholesale)')
    ap.add_argument('--fused', action='store_true', help='workload c5, one GPU: the one-launch resample + clip (resample_stack_sigclip) '
                    'instead of the two-step default')
    ap.add_argument('--no-gather', action='store_true', help="N > 1, --exchange rs: leave the mean ROW-DISTRIBUTED (no all-gather of the "
                    "result rows: 10 instead of 14 bytes per pixel on the wire); every rank keeps its rows of every stripe")
    ap.add_argument('--stripes', type=int, default=0, help='row stripes for collective/compute overlap (N > 1); 0 = by payload '
                                                            '(parallel.default_stripes: 4 for 4096 x 4096)')
    ap.add_argument('--hier-shards', type=int, default=8, help='--scaling strong: the job is this many shards of '
                                                               'total-frames / shards frames, whatever --gpus is')
    ap.add_argument('--exact-moments', action='store_true', help='APGPU_STACK_EXACT_MOMENTS: float64 clip only (no float32 fast path)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--force-collective', action='store_true', help='run the striped all-reduce path even with one rank (testing)')
    ap.add_argument('--cpu-seconds', type=float, default=12.0, help='target CPU time of each cpu_baseline leg')
    ap.add_argument('--master-port', type=int, default=0, help='rendezvous port of the built-in launcher (0 = pick a free one)')
    ap.add_argument('--selftest-cpu', action='store_true',
                    help='launcher/rendezvous/reporting self-test on CPU: gloo backend, a trivial stand-in step, no kernels '
                         '(tests/test_bench_launcher.py); the JSON line is marked "selftest": true and carries no roofline')
    args = ap.parse_args(argv)
    base = {'c2': (64, 4096, 4096), 'c4': (64, 6248, 4176), 'c5': (16, 8192, 8192)}[args.workload]
    args.baseline_dims = args.frames is None and args.height is None and args.width is None
    if args.frames is None:
        args.frames = base[0]
    if args.height is None:
        args.height = base[1]
    if args.width is None:
        args.width = base[2]
    return args


# -------------------------------------------------------------------------------------------------------------
# built-in launcher: N fresh child processes, one per GPU.  Runs before torch is imported in this process, so the
# parent never initialises the GPU (and nothing is exec'ed from a process that did).
# -------------------------------------------------------------------------------------------------------------
def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def launch_ranks(args, argv):
    port = args.master_port or _free_port()
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ)
        env.update(RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), LOCAL_WORLD_SIZE=str(args.gpus),
                   MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), APGPU_BENCH_CHILD='1')
        env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')       # dmabuf IPC (RCCL across processes on this pool)
        # rank 0 owns stdout (the JSON line); the other ranks' stdout goes to stderr
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env,
                                      stdout=None if r == 0 else sys.stderr))
    rc = 0
    deadline = None
    alive = list(procs)
    while alive:
        for p in list(alive):
            code = p.poll()
            if code is not None:
                alive.remove(p)
                if code != 0 and rc == 0:
                    rc = code
                    deadline = time.time() + 30.0               # one rank failed: give the others 30 s, then stop them
        if deadline is not None and time.time() > deadline:
            for p in alive:
                p.kill()                                        # exactly the PIDs started above
            break
        time.sleep(0.05)
    return rc if rc >= 0 else 1


def cpu_model():
    try:
        for ln in open('/proc/cpuinfo'):
            if ln.startswith('model name'):
                return ln.split(':', 1)[1].strip()
    except OSError:
        pass
    return 'unknown CPU'


def cpu_baseline(frames, masters, nflat, e, seconds):
    """Times the oracle's fused calibrate + clipped stack (oracle/apref.c, OpenMP: all threads, and one thread) on a
    row sample, and the NumPy path (oracle/numpy_ref.py: single process and multiprocessing) in a child process."""
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

    def leg(seconds):
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
        return rows, times[len(times) // 2], len(times), t_total

    model = cpu_model()
    rows, t, nruns, t_total = leg(seconds)
    out = dict(value=N * rows * W / 1e6 / t, unit='Mpixels/s', cores=threads, kind='port',
               sample='%d frames x %d rows x %d cols f32, median of %d runs (%.1f s of CPU work, OpenMP %d threads on %s, '
                      'oracle/apref.c fused calibrate + clipped stack)' % (N, rows, W, nruns, t_total, threads, model))
    if threads > 1:
        apref.set_num_threads(1)
        rows1, t1, nruns1, tt1 = leg(min(seconds, 6.0))
        apref.set_num_threads(threads)
        out['single_thread'] = dict(value=N * rows1 * W / 1e6 / t1, unit='Mpixels/s', cores=1,
                                    sample='%d rows, median of %d runs (%.1f s)' % (rows1, nruns1, tt1))
    # the NumPy path the north star names, in a fresh child process (it forks workers; this process holds the GPU)
    try:
        r = subprocess.run([sys.executable, os.path.join(ROOT, 'oracle', 'numpy_ref.py'), '--frames', str(N), '--width', str(W),
                            '--seconds', str(min(seconds, 10.0))], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, timeout=240)
        nr = json.loads(r.stdout.decode().strip().splitlines()[-1])
        out['numpy'] = dict(single_process=nr['single'], multiprocessing=nr.get('multi'), cpu=nr['cpu'], numpy=nr['numpy'],
                            kind='port',
                            sample='oracle/numpy_ref.py: ApCalibrate.py:439-464 per frame + sigma_clipped_stats(axis=0) in NumPy '
                                   '(nanmedian/nanstd along N), %d-frame row blocks' % N)
    except Exception as exc:                                  # the baseline is a report, never a reason to lose the bench line
        out['numpy'] = dict(error=repr(exc))
    return out


def selftest_cpu(args, world, rank):
    """Launcher / rendezvous / timing / reporting on CPU (gloo): no kernels, a stand-in step."""
    import torch
    import torch.distributed as dist
    sys.stdout.flush()
    saved_stdout = os.dup(1)                                 # gloo chats on the C-level stdout: keep it for the JSON line
    os.dup2(2, 1)
    if world > 1:
        dist.init_process_group('gloo', rank=rank, world_size=world)
    x = torch.full((256, 256), float(rank + 1))
    t0 = time.perf_counter()
    for _ in range(args.steps):
        y = (x * 2).sum()
    elapsed = time.perf_counter() - t0
    per_rank = [elapsed]
    rworld = 1
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64)
        lst = [torch.zeros(1, dtype=torch.float64) for _ in range(world)]
        dist.all_gather(lst, t)
        per_rank = [float(v.item()) for v in lst]
        tot = torch.tensor([float(y)])
        dist.all_reduce(tot)
        rworld = dist.get_world_size()
        assert float(tot) == sum(2.0 * 65536 * (r + 1) for r in range(world))
        dist.barrier()
        dist.destroy_process_group()
    if rworld != args.gpus:
        print('error: world size %d != --gpus %d' % (rworld, args.gpus), file=sys.stderr)
        return 3
    if rank == 0:
        os.write(saved_stdout, (json.dumps({'selftest': True, 'n_gpus': world, 'rccl_world_size': rworld, 'steps': args.steps,
                                            'per_rank_ms': [1e3 * t / args.steps for t in per_rank], 'backend': 'gloo'}) + '\n').encode())
    return 0


def main(argv=None):
    argv = sys.argv[1:] if argv is None else argv
    args = parse(argv)
    under_launcher = 'WORLD_SIZE' in os.environ
    if args.gpus > 1 and not under_launcher:
        return launch_ranks(args, argv)
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if world != args.gpus:
        print('error: --gpus %d but the launcher started WORLD_SIZE=%d ranks' % (args.gpus, world), file=sys.stderr)
        return 3
    if args.selftest_cpu:
        return selftest_cpu(args, world, rank)

    # Only the JSON line may appear on stdout: RCCL prints a version banner to the C-level stdout at
    # exit, so everything else (C and Python) is routed to stderr until the final print.
    sys.stdout.flush()
    saved_stdout = os.dup(1)
    os.dup2(2, 1)

    import torch
    import torch.distributed as dist

    if world > 1 or args.force_collective:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29517')
        os.environ.setdefault('RANK', '0')
        os.environ.setdefault('WORLD_SIZE', '1')
        if os.environ.get('APGPU_BENCH_ONE_GPU_TEST'):
            # test hook (tests/test_gpu_bench_contract.py): every rank uses GPU 0 and the ranks talk over gloo, so that the
            # whole N > 1 code path - sharding, striped exchange, hierarchical shards, the row-shard leg, the reporting -
            # runs with real kernels on a one-GPU box.  The numbers of such a run mean nothing and the line says so.
            local_rank = 0
            dist.init_process_group('gloo')
        else:
            dist.init_process_group('nccl', device_id=torch.device('cuda', local_rank))
    rccl_world = dist.get_world_size() if dist.is_initialized() else 1
    if rccl_world != args.gpus:
        print('error: process group has %d ranks, --gpus %d' % (rccl_world, args.gpus), file=sys.stderr)
        return 3
    torch.cuda.set_device(local_rank)
    dev = torch.device('cuda', local_rank)

    from astrophotography_amd import ops, synth, parallel

    wl = args.workload
    rowshard = args.parallelism == 'rowshard'
    strong = args.scaling == 'strong'
    H_glob, W = args.height, args.width
    if rowshard:
        # every rank: all N frames of its row block.  weak: H rows per rank (a world*H-row mosaic), strong: H/world rows
        N = args.total_frames if strong else args.frames
        if strong:
            r0, r1 = parallel.row_block(H_glob, world, rank)
            H = r1 - r0
        else:
            H = H_glob
        n_total = N
    else:
        if strong:
            f0, f1 = parallel.shard_frames(args.total_frames, world, rank)
            N = f1 - f0
            n_total = args.total_frames
        else:
            N = args.frames
            f0 = rank * N
            n_total = world * N
        H = H_glob
    P = H * W
    if wl != 'c2' and (strong or rowshard):
        print('error: --scaling strong / --parallelism rowshard are defined for workload c2', file=sys.stderr)
        return 2
    masters = synth.make_masters(H, W, config_id=2, device=dev)
    nflat, _ = ops.flat_normalize(masters['flat'])
    tdtype = torch.float32 if args.dtype == 'f32' else torch.uint16
    frames = None
    c5_affines = None
    if wl == 'c5':
        import numpy as np
        rng = np.random.default_rng(5000 + rank)
        th = np.deg2rad(rng.uniform(-0.2, 0.2, N))
        c5_affines = np.stack([np.cos(th), -np.sin(th), rng.uniform(-3, 3, N), np.sin(th), np.cos(th), rng.uniform(-3, 3, N)], 1)
    if wl != 'c4':
        first = (rank * 1000) if rowshard else f0               # distinct synthetic frames per rank
        scenes = None
        if wl == 'c5' and args.c5_dithered:
            # frame f sees scene(A_f^-1 q): the scene resampled with the inverse transform (this library's own Lanczos-3 warp;
            # pixels the warp leaves undefined get the sky level)
            def scenes(f):
                a = c5_affines[f]
                M = np.array([[a[0], a[1], a[2]], [a[3], a[4], a[5]], [0.0, 0.0, 1.0]])
                Mi = np.linalg.inv(M)
                w, _ = ops.resample_affine(masters['scene'][None], [[Mi[0, 0], Mi[0, 1], Mi[0, 2], Mi[1, 0], Mi[1, 1], Mi[1, 2]]], weight=False)
                return torch.nan_to_num(w[0], nan=525.0)
        frames = synth.make_frames(N, masters, nflat, config_id=2, dtype=tdtype, first_frame=first, scenes=scenes)
    e = synth.EXP_RATIO
    calib = dict(bias=masters['bias'], dark=masters['dark'], nflat=nflat,
                 exp_ratio=torch.full((N,), e, dtype=torch.float32, device=dev), dark_still_biased=False)
    torch.cuda.synchronize()

    if wl == 'c4':
        if world > 1:
            print('error: workload c4 is a single-GPU configuration', file=sys.stderr)
            return 2
        nflat4, _ = ops.bayer_flat_normalize(masters['flat'])
        frames = synth.make_frames(N, masters, nflat4, config_id=4, dtype=torch.uint16, first_frame=0)
        calib = dict(calib, nflat=nflat4)
    if wl == 'c5':
        affines = c5_affines
        cal = ops.calibrate(frames, masters['bias'], masters['dark'], nflat, e)
        del frames
        frames = cal
        st = ops.sigclip_global(masters['dark'], sigma=4.0, maxiters=5)
        badmask, _ = ops.threshold_mask(masters['dark'], thresholds=st[3:5].contiguous())     # lo, hi of the clip, read on the device
        resampled = torch.empty_like(frames)
    torch.cuda.synchronize()

    hier_chunk = None
    if strong and not rowshard:
        if args.total_frames % args.hier_shards or args.hier_shards % world:
            print('error: --total-frames must split into --hier-shards equal shards, and the shards over the ranks', file=sys.stderr)
            return 2
        hier_chunk = args.total_frames // args.hier_shards
    single_launch = (world == 1 and not args.force_collective and hier_chunk is None) or rowshard
    payload = 'f64i' if args.exchange == 'rs' else args.exchange     # the moment layout the kernels write ('rs': float64 sum + int32 count)
    # 'rs' sends the count as a float16 plane while that is exact (<= 2048 frames in the whole job), else as int32
    count_dtype = torch.float16 if (args.rs_count == 'f16' or (args.rs_count == 'auto' and n_total <= 2048)) else torch.int32
    count_bytes = 2 if count_dtype == torch.float16 else 4
    n_stripes = args.stripes or parallel.default_stripes(H, W, payload)
    timings = []                                             # (start, end) events around every stripe's all-reduce

    def step():
        if wl == 'c4':
            return ops.stack_median(frames, calib=calib)
        if wl == 'c5':
            if args.fused and world == 1:
                return ops.resample_stack_sigclip(frames, affines, mask=badmask, sigma=3.0, maxiters=5, outputs=('mean',))['mean']
            ops.resample_affine(frames, affines, mask=badmask, out=resampled, weight=False)
            if world == 1:
                return ops.stack_sigclip(resampled, sigma=3.0, maxiters=5, outputs=('mean',))['mean']
            return parallel.stack_nshard(resampled, None, sigma=3.0, maxiters=5, n_stripes=n_stripes, exchange=args.exchange,
                                         timings=timings, count_dtype=count_dtype)
        if single_launch:
            return ops.stack_sigclip(frames, sigma=3.0, maxiters=5, cenfunc='median', stdfunc='std', calib=calib,
                                     outputs=('mean',), exact=args.exact_moments)['mean']
        return parallel.stack_nshard(frames, calib, sigma=3.0, maxiters=5, n_stripes=n_stripes,
                                     force_collective=args.force_collective, exchange=args.exchange, hier_chunk=hier_chunk,
                                     timings=timings, count_dtype=count_dtype, gather=not (args.no_gather and args.exchange == 'rs'))

    for _ in range(args.warmup):
        out = step()
    torch.cuda.synchronize()
    ops.stack_redo_stats(reset=True)                         # count the timed steps only
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()

    del timings[:]
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    t0 = time.perf_counter()
    for k in range(args.steps):
        evs[k][0].record()
        out = step()
        evs[k][1].record()
    torch.cuda.synchronize()
    local_elapsed = time.perf_counter() - t0                    # this rank's own time for the K steps
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    per_rank_ms = [1e3 * local_elapsed / args.steps]
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        mine = torch.tensor([local_elapsed], dtype=torch.float64, device=dev)
        allt = [torch.empty(1, dtype=torch.float64, device=dev) for _ in range(world)]
        dist.all_gather(allt, mine)
        per_rank_ms = [1e3 * float(v.item()) / args.steps for v in allt]
    ms_per_step = 1e3 * elapsed / args.steps
    # what the stack's fast kernels left to the redo pass over the timed steps (counters in the workspace, read after the clock)
    redo = ops.stack_redo_stats()
    job_pixels = float(N) * P                                  # input frame pixels of ALL ranks (ragged blocks: summed, not
    if world > 1:                                              # rank 0's share times the world size)
        t = torch.tensor([job_pixels], dtype=torch.float64, device=dev)
        dist.all_reduce(t)
        job_pixels = float(t.item())
    value = job_pixels / 1e6 / (elapsed / args.steps)

    def timed_extra(fn, steps):
        """Mean wall time per step (ms, max over ranks) of `fn`, bracketed like the main loop."""
        fn()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        ta = time.perf_counter()
        for _ in range(steps):
            fn()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        dt = torch.tensor([time.perf_counter() - ta], dtype=torch.float64, device=dev)
        if world > 1:
            dist.all_reduce(dt, op=dist.ReduceOp.MAX)
        return 1e3 * float(dt.item()) / steps

    # the all-reduces of the timed steps on the communication stream (per step: summed over the stripes; max over ranks)
    exchange_ms = None
    if timings:
        ex = sum(a.elapsed_time(b) for a, b in timings) / args.steps
        t = torch.tensor([ex], dtype=torch.float64, device=dev)
        if world > 1:
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        exchange_ms = float(t.item())

    extra = {}
    if wl == 'c2' and not rowshard and not single_launch:
        # (0) what one SCALE run needs to diagnose itself: the stripes' moment kernels alone (no collective: `compute_ms`, max over
        # ranks), the collectives' time on the communication stream (`exchange_ms` above) and how much of it hid behind the
        # kernels: overlap = (compute + exchange - step) / exchange, 1 = fully hidden, 0 = fully exposed
        import functools
        lm = functools.partial(parallel._default_local_moments, want_std=False, hier_chunk=hier_chunk, exact=False)
        clipkw = dict(sigma=3.0, maxiters=5, cenfunc='median', stdfunc='std')
        rows_of = parallel.stripe_rows(H, n_stripes if (world > 1 or args.force_collective) else 1)
        cms = timed_extra(lambda: [lm(frames, calib, a, b, clipkw, payload) for a, b in rows_of], min(args.steps, 10))
        extra['compute_ms'] = cms
        if exchange_ms:
            extra['exchange_overlap_frac'] = max(0.0, min(1.0, (cms + exchange_ms - ms_per_step) / exchange_ms))
            extra['exposed_exchange_ms'] = max(0.0, ms_per_step - cms)
        # (1) strong scaling at one rank: the exact kernel on all frames beside the hierarchical reduction
        if strong and world == 1:
            ems = timed_extra(lambda: ops.stack_sigclip(frames, sigma=3.0, maxiters=5, calib=calib, outputs=('mean',)), min(args.steps, 5))
            extra['exact_ms'] = ems
            extra['exact_value'] = job_pixels / 1e6 / (ems * 1e-3)
            extra['exact_kernel'] = ops.stack_kernel_name(N, args.dtype, calibrated=True, outputs=('mean',))
        # (2) the row-shard form of the same job (exact at every N, no data-path collective), timed beside the N-shard one
        if world > 1:
            if strong:
                r0, r1 = parallel.row_block(H, world, rank)
                del frames
                rmasters = {k: v[r0:r1].contiguous() for k, v in masters.items()}
                rnflat = nflat[r0:r1].contiguous()
                rframes = synth.make_frames(n_total, rmasters, rnflat, config_id=2, dtype=tdtype, first_frame=0)
                rcal = dict(bias=rmasters['bias'], dark=rmasters['dark'], nflat=rnflat,
                            exp_ratio=torch.full((n_total,), e, dtype=torch.float32, device=dev), dark_still_biased=False)
                rpix = float(n_total) * (r1 - r0) * W
            else:
                rframes, rcal, rpix = frames, calib, float(N) * P
            rms = timed_extra(lambda: parallel.stack_rowshard(rframes, rcal, sigma=3.0, maxiters=5, outputs=('mean',)), args.steps)
            t = torch.tensor([rpix], dtype=torch.float64, device=dev)
            dist.all_reduce(t)
            extra['rowshard'] = {'ms_per_step': rms, 'value': float(t.item()) / 1e6 / (rms * 1e-3), 'unit': 'Mpixels/s',
                                 'kernel': ops.stack_kernel_name(rframes.shape[0], args.dtype, calibrated=True, outputs=('mean',)),
                                 'note': 'every rank: all frames of its own rows, exact clip, no data-path collective'}
            if strong:
                frames = rframes

    # dominant kernel: device time per step from HIP events on the launch stream; for a single-launch step this is
    # exactly one launch of the stack kernel, whose name the library reports for the variant it dispatched.
    kern_ms = sorted(a.elapsed_time(b) for a, b in evs)
    avg_kernel_ms = sum(kern_ms) / len(kern_ms)
    esize = 4 if args.dtype == 'f32' else 2
    nshard_multi = not single_launch
    out_bytes = 4 if single_launch else {'f64': 24, 'f64i': 20, 'f32': 12}[payload]           # mean plane | moment planes written
    algo_bytes = esize * N * P + 12 * P + out_bytes * P       # frames + bias/dark/nflat read, outputs written
    kernel_name = ops.stack_kernel_name(min(N, hier_chunk) if hier_chunk else N, args.dtype, calibrated=True,
                                        outputs=('mean',) if single_launch else ({'f64': ('moments_f64p',), 'f64i': ('moments_f64',), 'f32': ('moments',)}[payload]),
                                        moments_mean_only=not single_launch and payload != 'f32', exact=args.exact_moments and single_launch)
    metric = 'Mpixels/sec calibrate+sigma-clip-stack'
    cfg_name = 'C3' if strong else 'C2'
    if not strong and not (N == 64 and H_glob == 4096 and W == 4096 and args.dtype == 'f32'):
        cfg_name = 'C2-like (not BASELINE\'s 64x4096x4096 f32)'
    workload = '%s: %dx%dx%d %s per GPU, fused bias/dark/flat + 3-sigma maxiters-5 median-centred clipped mean' % (
        cfg_name, N, H, W, args.dtype)
    if wl == 'c4':
        algo_bytes = 2 * N * P + 12 * P + 4 * P
        args.dtype = 'u16'
        kernel_name = ops.stack_kernel_name(N, 'u16', calibrated=True, median_only=True)
        metric = 'Mpixels/sec calibrate+median-stack (uint16 Bayer)'
        workload = '%s: %dx%dx%d u16 RGGB mosaic, per-channel flat normalisation, fused bias/dark/flat + median stack' % (
            'C4' if (N, H, W) == (64, 6248, 4176) else 'C4-like (not BASELINE\'s 64x6248x4176)', N, H, W)
    if wl == 'c5':
        algo_bytes = (8 * N * P + P) + (4 * N * P + out_bytes * P)     # resample read+write (+mask), stack read + outputs
        kernel_name = 'resample_affine_kernel + ' + ops.stack_kernel_name(N, 'f32', calibrated=False) + ' (step = both launches)'
        if args.fused and world == 1:
            algo_bytes = 4 * N * P + P + 4 * P + out_bytes * P           # frames + mask read, the hit-bit plane written and read, mean written
            kernel_name = 'resample_clip_kernel_v2<%d> (one launch + tile records + bad-pixel bits)' % (4 * ((N + 3) // 4))
        metric = 'Mpixels/sec mask+affine-resample+sigma-clip-stack'
        workload = ('%s: %dx%dx%d f32 calibrated frames, bad-pixel mask, per-frame affine Lanczos-3 resample, 3-sigma maxiters-5 clipped mean' + (
            '; DITHERED synthetic sequence (the transforms align the stars)' if args.c5_dithered else '')) % (
            'C5 (per-GPU share of 128x8192x8192 on 8 GPUs)' if (N, H, W) == (16, 8192, 8192) else 'C5-like (not BASELINE\'s 16x8192x8192 share)', N, H, W)
    achieved = algo_bytes / (avg_kernel_ms * 1e-3) / 1e9

    # HBM traffic per launch from the PMC counters: collected by profiles/run_profile.sh (separate --pmc passes of THIS
    # command under rocprofv3; counters cannot be read from inside the process) and looked up by workload key.
    traffic = None
    traffic_source = None
    valu = None                                              # the second bound: VALU issue (SQ counters of the same command)
    tkey = '%s:%dx%dx%d:%s:%s' % (wl, N, H, W, args.dtype, 'single' if single_launch else args.exchange)
    tfile = os.path.join(ROOT, 'profiles', 'pmc_traffic.json')
    if os.path.exists(tfile) and world == 1:
        try:
            td = json.load(open(tfile))
            ent = td.get('workloads', {}).get(tkey)
            # (per LAUNCH like `achieved`: only a step that IS one launch of that kernel reports it)
            if ent and ent.get('kernel') == kernel_name and single_launch:   # counters of ANOTHER kernel build are not this kernel's traffic
                traffic = ent.get('hbm_bytes_per_launch')
                traffic_source = 'profiles/pmc_traffic.json (%s)' % ent.get('tag')
                if ent.get('valu'):
                    valu = dict(ent['valu'], source='profiles/pmc_traffic.json (%s): SQ_INSTS_VALU / SQ_WAVES, SQ_ACTIVE_INST_VALU x 4 / '
                                                   '(SIMDs x GRBM_GUI_ACTIVE / XCDs), GRBM_GUI_ACTIVE / XCDs / duration' % ent.get('tag'))
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

    line = None
    if rank == 0:
        if world == 1:
            par = 'single GPU'
        elif rowshard:
            par = 'row-shard x%d (every rank: all %d frames of %d rows; no data-path collective)' % (world, N, H)
        else:
            par = 'N-shard x%d (%d of %d frames per rank%s), %d stripes, ONE all-reduce per stripe of %s' % (
                world, N, n_total, ', clipped in shards of %d' % hier_chunk if hier_chunk else '', n_stripes,
                'float64 sum + count (16 B/pixel)' if args.exchange == 'f64' else 'float32 sum + count (8 B/pixel)')
            if args.exchange == 'rs':
                par = 'N-shard x%d (%d of %d frames per rank%s), %d stripes, per stripe: reduce-scatter by rows of the float64 sum plane and the %s count plane, every rank finalises its rows, all-gather of the float32 mean rows' % (
                    world, N, n_total, ', clipped in shards of %d' % hier_chunk if hier_chunk else '', n_stripes, 'float16' if count_bytes == 2 else 'int32')
        if world == 1 and hier_chunk:
            par = 'single GPU, hierarchical: %d shards of %d frames clipped per shard, float64 moments added' % (N // hier_chunk, hier_chunk)
        line = {
            'metric': metric, 'value': value, 'unit': 'Mpixels/s',
            'n_gpus': world, 'rccl_world_size': rccl_world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': ms_per_step,
            'per_rank_ms': per_rank_ms,
            **({'one_gpu_test': True} if os.environ.get('APGPU_BENCH_ONE_GPU_TEST') else {}),
            'higher_is_better': True, 'scaling': args.scaling, 'vs_baseline': None, 'dtype': args.dtype, 'data': 'synthetic',
            'config': {'workload': workload, 'frames_per_gpu': N, 'frames_total': n_total if not rowshard else N,
                       'height': H, 'width': W, 'parallelism': par},
            # share of the output pixels of the timed steps' stack calls that did not finish on the fast kernel (listed pixels +
            # 64 x blocks given up, over pixels; rank 0's calls) - the data-dependent part of the two-kernel scheme's cost
            'redo_fraction': redo['fraction'] if redo['calls'] else None,
            'redo': redo if redo['calls'] else None,
            # `bound`: the roofline the kernel is PRICED against (bytes / HBM peak, the contract's fields); `limiter`: what the
            # counters say holds it back today - the fused float32 clip kernels issue VALU instructions ~90 % of the time
            # (roofline.valu, measured), the uint16 median kernel streams.
            'roofline': {'bound': 'hbm', 'limiter': ('valu_issue' if (valu and valu.get('busy_frac', 0) > 0.8) else
                                                     ('hbm' if wl == 'c4' else 'valu_issue (not re-measured for this key)')),
                         'achieved': achieved, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                         'frac': achieved / HBM_PEAK_GBS, 'traffic': traffic, 'traffic_source': traffic_source, 'traffic_key': tkey,
                         'valu': valu, 'kernel': kernel_name,
                         'avg_launch_ms': avg_kernel_ms, 'min_launch_ms': kern_ms[0], 'algorithmic_bytes': algo_bytes,
                         'measured_copy_GBps': copy_gbs, 'frac_of_measured_copy': achieved / copy_gbs if copy_gbs else None},
        }
        if nshard_multi:
            line['roofline']['note'] = ('step = %d stripe kernels%s + all-reduces on side streams; achieved = per-rank algorithmic '
                                        'bytes / step time on the launch stream' % (
                                            n_stripes if world > 1 or args.force_collective else 1,
                                            ' x %d shards' % (N // hier_chunk) if hier_chunk and N > hier_chunk else ''))
            line['exchange'] = args.exchange
            line['exchange_bytes_per_pixel'] = parallel.exchange_bytes_per_pixel(payload, count_bytes=count_bytes)
            line['exchange_bytes_on_wire'] = parallel.exchange_bytes_on_wire(args.exchange, world, P, count_bytes=count_bytes,
                                                                             gather=not (args.no_gather and args.exchange == 'rs'))
            line['exchange_gathered'] = not (args.no_gather and args.exchange == 'rs')
            if args.exchange == 'rs':
                line['exchange_count_dtype'] = 'float16' if count_bytes == 2 else 'int32'
            line['exchange_ms'] = exchange_ms
            line['stripes'] = n_stripes
            if hier_chunk:
                line['hier_shards'] = args.hier_shards
        line.update(extra)
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
    return 0


if __name__ == '__main__':
    sys.exit(main())
