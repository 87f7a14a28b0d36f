#!/usr/bin/env python3
"""Per-kernel throughput of the whole path on one MI355X (development/evidence aid, not the contract
bench): algorithmic GB/s of every entry point of libapgpu.so against the 8 TB/s HBM peak."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from astrophotography_amd import ops, synth


def timeit(fn, reps=10, warm=2, batch=1):
    """Median / min device time per call.  batch > 1 queues that many calls between the two events so
    that sub-100-us kernels are not timed through the host's launch gaps."""
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(batch):
            fn()
        b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) / batch)
    ts.sort()
    return ts[len(ts) // 2], ts[0]


def cpu_side(frames, masters, nflat, calib, rows):
    """The oracle (the CPU restatement of the reference arithmetic, oracle/apref.c; OpenMP where it has it) on a
    bounded sample of the same data, beside the GPU numbers: Mpixels/s of input frame-pixels for the slab kernels,
    of image pixels for the image kernels."""
    import numpy as np
    from oracle import apref
    thr = apref.num_threads()
    rowsN = 256                                                  # sample: 64 frames x 256 rows x 4096 columns
    raw = frames[:, :rowsN].cpu().numpy()
    b, d, nf = masters['bias'][:rowsN].cpu().numpy(), masters['dark'][:rowsN].cpu().numpy(), nflat[:rowsN].cpu().numpy()
    img = frames[0].cpu().numpy()
    dark = masters['dark'].cpu().numpy()
    bias_full = masters['bias'].cpu().numpy()

    def t(fn, reps=3):
        fn()
        ts = []
        for _ in range(reps):
            t0 = time.perf_counter()
            fn()
            ts.append(time.perf_counter() - t0)
        return sorted(ts)[len(ts) // 2]

    def out(name, npix, sec, gpu_name):
        g = next((r for r in rows if r['kernel'].startswith(gpu_name)), None)
        gpu_rate = None
        if g is not None:
            gpu_pix = {True: 64 * 4096 * 4096, False: 4096 * 4096}[gpu_name.startswith(('calibrate', 'stack', 'resample'))]
            gpu_rate = gpu_pix / (g['ms'] * 1e-3) / 1e6
        rows.append(dict(kernel='CPU oracle: ' + name, cpu_Mpix_s=npix / sec / 1e6, threads=thr, gpu_Mpix_s=gpu_rate))
        print('CPU %-44s %10.1f Mpix/s (%d threads)   GPU %s Mpix/s' % (name, npix / sec / 1e6, thr,
              '%.0f' % gpu_rate if gpu_rate else '-'), flush=True)

    n_in = raw.size
    out('calibrate (A2)', n_in, t(lambda: apref.calibrate(raw, b, d, nf, synth.EXP_RATIO)), 'calibrate')
    out('fused calibrate + clipped mean (A2+A7)', n_in, t(lambda: apref.calibrate_stack(raw, b, d, nf, synth.EXP_RATIO)), 'stack_sigclip fused, mean (A2')
    cal = apref.calibrate(raw, b, d, nf, synth.EXP_RATIO)
    out('clipped mean/median/std planes (A7)', n_in, t(lambda: apref.stack_sigclip(cal, sigma=3.0, maxiters=5)), 'stack_sigclip fused, mean+median')
    out('ccdproc configuration (A6)', n_in, t(lambda: apref.combine_ccdproc(cal.astype(np.float64), 5.0, 5.0), reps=1), 'stack ccdproc')
    out('median stack', n_in, t(lambda: apref.stack_median(cal)), 'stack_median fused (C4')
    out('flat_normalize (A1)', img.size, t(lambda: apref.flat_normalize(masters['flat'].cpu().numpy())), 'flat_normalize')
    out('sigclip_global (A3)', img.size, t(lambda: apref.sigclip_global(dark, sigma=4.0, maxiters=5), reps=1), 'sigclip_global')
    st = apref.sigclip_global(dark, sigma=4.0, maxiters=5)
    lo, hi = apref.badpix_thresholds(st['median'], st['std'], 4.0)
    out('threshold_mask (A4)', img.size, t(lambda: apref.threshold_mask(dark, lo, hi)), 'threshold_mask')
    m = apref.threshold_mask(dark, 5.0, 35.0)[0]
    out('fix_badpix delta 2 (A5)', img.size, t(lambda: apref.fix_badpix(img, m, 2)), 'fix_badpix')
    out('imarith SUB (A8)', img.size, t(lambda: apref.imarith(img, 'SUB', bias_full)), 'imarith')
    rng = np.random.default_rng(5)
    th = np.deg2rad(rng.uniform(-0.2, 0.2, 4))
    A = np.stack([np.cos(th), -np.sin(th), rng.uniform(-3, 3, 4), np.sin(th), np.cos(th), rng.uniform(-3, 3, 4)], 1)
    f4 = frames[:4].cpu().numpy()
    out('resample_affine Lanczos-3 (F3)', f4.size, t(lambda: apref.resample_affine(f4, A), reps=1), 'resample_affine 64')


def main():
    H = W = 4096
    P = H * W
    N = 64
    masters = synth.make_masters(H, W, config_id=2, device='cuda')
    nflat, _ = ops.flat_normalize(masters['flat'])
    frames = synth.make_frames(N, masters, nflat, config_id=2)
    calib = dict(bias=masters['bias'], dark=masters['dark'], nflat=nflat, exp_ratio=synth.EXP_RATIO)
    rows = []

    only = sys.argv[1] if len(sys.argv) > 1 and not sys.argv[1].startswith('--') else ''

    def rec(name, bytes_, fn, **kw):
        if only and only not in name:
            return
        med, best = timeit(fn, **kw)
        rows.append(dict(kernel=name, ms=med, ms_min=best, GBps=bytes_ / med / 1e6, frac_of_8TBps=bytes_ / med / 1e6 / 8000))
        print('%-46s %8.3f ms  %7.0f GB/s  %5.1f %%' % (name, med, bytes_ / med / 1e6, 100 * bytes_ / med / 1e6 / 8000), flush=True)

    out = torch.empty_like(frames)
    rec('calibrate f32 slab 64x4096^2 (A2)', (4 + 4) * N * P + 12 * P, lambda: ops.calibrate(frames, masters['bias'], masters['dark'], nflat, synth.EXP_RATIO, out=out))
    del out
    rec('stack_sigclip fused, mean (A2+A7, bench kernel)', 4 * N * P + 16 * P, lambda: ops.stack_sigclip(frames, calib=calib, outputs=('mean',)))
    rec('stack_sigclip fused, moments (N-shard partials)', 4 * N * P + 24 * P, lambda: ops.stack_sigclip(frames, calib=calib, outputs=('moments',)))
    rec('stack_sigclip fused, mean+median+std (EXTRA)', 4 * N * P + 24 * P, lambda: ops.stack_sigclip(frames, calib=calib, outputs=('mean', 'median', 'std')), reps=5)
    rec('stack_sigclip plain f32 (A7, no calibration)', 4 * N * P + 4 * P, lambda: ops.stack_sigclip(frames, outputs=('mean',)))
    rec('stack ccdproc config: 1 pass, median/mad_std 5s (A6)', 4 * N * P + 4 * P, lambda: ops.stack_sigclip(frames, sigma=5.0, maxiters=1, stdfunc='mad_std', outputs=('mean',)), reps=3, warm=1)
    # what ApMasterCal.make_master really runs (ap_combine_darks.py:394-420): the frames as stored - float32, or raw uint16 darks /
    # biases -, no calibration, one pass median / mad_std 5 sigma, float64 mean + std planes + count (bytes: frames + 20 per pixel)
    rec('ApMasterCal A6: f32 frames -> mean_f64, std_f64, count', 4 * N * P + 20 * P,
        lambda: ops.stack_sigclip(frames, sigma=5.0, maxiters=1, stdfunc='mad_std', outputs=('mean_f64', 'count', 'std_f64')), reps=3, warm=1)
    raw16 = synth.make_frames(N, masters, nflat, config_id=2, dtype=torch.uint16)
    rec('ApMasterCal A6: uint16 frames -> mean_f64, std_f64, count', 2 * N * P + 20 * P,
        lambda: ops.stack_sigclip(raw16, sigma=5.0, maxiters=1, stdfunc='mad_std', outputs=('mean_f64', 'count', 'std_f64')), reps=3, warm=1)
    del raw16
    rec('stack_sigclip fused, 48 of 64 slots (non-FULL lean)', 4 * 48 * P + 16 * P, lambda: ops.stack_sigclip(frames[:48], calib=calib, outputs=('mean',)))
    rec('stack_median fused (C4-style, f32)', 4 * N * P + 16 * P, lambda: ops.stack_median(frames, calib=calib))
    f16 = synth.make_frames(N, masters, nflat, config_id=2, dtype=torch.uint16)
    rec('stack_sigclip fused u16 raw', 2 * N * P + 16 * P, lambda: ops.stack_sigclip(f16, calib=calib, outputs=('mean',)))
    rec('stack_median fused u16 raw (C4-style)', 2 * N * P + 16 * P, lambda: ops.stack_median(f16, calib=calib))
    del f16
    img = frames[0].contiguous()
    rec('flat_normalize 4096^2 (A1)', 12 * P, lambda: ops.flat_normalize(masters['flat']), batch=20)
    rec('sigclip_global f32 4096^2, sigma 4, 5 iters (A3)', 4 * P * 10 * 3, lambda: ops.sigclip_global(masters['dark'], sigma=4.0, maxiters=5), reps=5)
    rec('threshold_mask 4096^2 (A4)', 5 * P, lambda: ops.threshold_mask(masters['dark'], 5.0, 35.0), batch=20)
    mask, _ = ops.threshold_mask(masters['dark'], 5.0, 35.0)
    rec('fix_badpix 4096^2, delta 2 (A5)', 9 * P, lambda: ops.fix_badpix(img, mask, 2), batch=20)
    print('bad pixel fraction %.4f' % float(mask.float().mean()))
    rec('imarith f32 SUB image 4096^2 (A8)', 12 * P, lambda: ops.imarith(img, 'SUB', masters['bias']), batch=20)
    import numpy as np
    rng = np.random.default_rng(5)
    th = np.deg2rad(rng.uniform(-0.2, 0.2, N))
    A = np.stack([np.cos(th), -np.sin(th), rng.uniform(-3, 3, N), np.sin(th), np.cos(th), rng.uniform(-3, 3, N)], 1)
    out = torch.empty_like(frames)
    rec('resample_affine 64x4096^2 f32, Lanczos-3 (F3)', 8 * N * P, lambda: ops.resample_affine(frames, A, out=out, weight=False), reps=5)
    rec('resample_affine + uint8 weight planes', 9 * N * P, lambda: ops.resample_affine(frames, A, out=out, weight=True), reps=5)
    del out
    # F4: the per-frame steps of calibrate_all.sh after the calibration (sky background mesh, cosmic rays); wall time of the
    # whole call (kernels + the small host steps on the mesh), quoted against one read + one write of the image
    cal, cr_truth = synth.make_sky_frame(H, W)                  # calibrated frame in electrons: sky, stars, noise, cosmic rays
    above = (cal > float(cal.median()) + 3 * float(cal.std())).to(torch.uint8)
    srcmask, _ = ops.source_mask(above, 5, 11)
    rec('source_mask 4096^2: labels >= 5 px + 11x11 dilation (F4)', 2 * P, lambda: ops.source_mask(above, 5, 11), reps=5)
    rec('box_clipped_stats 4096^2, 258x258 boxes, 3 sigma (F4)', 5 * P, lambda: ops.box_clipped_stats(cal, srcmask, 258, 258, sigma=3.0, maxiters=5), reps=5)
    rec('box_clipped_stats 4096^2, 128x128 boxes (LDS-resident)', 5 * P, lambda: ops.box_clipped_stats(cal, srcmask, 128, 128, sigma=3.0, maxiters=5), reps=5)
    from astrophotography_amd.core.ApMeasureBackground import ApMeasureBackground
    mb = ApMeasureBackground('ERROR')
    rec('ApMeasureBackground.process_data 4096^2 (F4, wall incl. host mesh steps)', 12 * P, lambda: mb.process_data(cal), reps=3, warm=1)
    r = ops.lacosmic(cal, None, niter=4)
    print('lacosmic: %d iterations, %d pixels flagged, %d of %d injected hits found' % (
        r[2], int(r[1].sum()), int((r[1].bool() & cr_truth).sum()), int(cr_truth.sum())))
    rec('lacosmic 4096^2, up to 4 iterations (F4)', 8 * P, lambda: ops.lacosmic(cal, None, niter=4), reps=3, warm=1)
    if '--cpu' in sys.argv:
        cpu_side(frames, masters, nflat, calib, rows)
    json.dump(rows, open(os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'gpurun_out', 'bench_kernels.json'), 'w'), indent=1)


if __name__ == '__main__':
    main()
