"""Development aid: where the single-rank time of the striped N-shard step goes (kernel with moment outputs, stripes, finalize)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from astrophotography_amd import ops, synth, parallel

N, H, W = 64, 4096, 4096
dev = torch.device('cuda', 0)
masters = synth.make_masters(H, W, config_id=2, device=dev)
nflat, _ = ops.flat_normalize(masters['flat'])
frames = synth.make_frames(N, masters, nflat, config_id=2)
calib = dict(bias=masters['bias'], dark=masters['dark'], nflat=nflat, exp_ratio=torch.full((N,), synth.EXP_RATIO, device=dev), dark_still_biased=False)


def t(fn, n=20):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n


print('mean                      %.3f ms' % t(lambda: ops.stack_sigclip(frames, calib=calib, outputs=('mean',))))
print('mean exact                %.3f ms' % t(lambda: ops.stack_sigclip(frames, calib=calib, outputs=('mean',), exact=True)))
print('moments f32               %.3f ms' % t(lambda: ops.stack_sigclip(frames, calib=calib, outputs=('moments',))))
print('moments_f64p mean-only    %.3f ms' % t(lambda: ops.stack_sigclip(frames, calib=calib, outputs=('moments_f64p',), moments_mean_only=True)))
print('moments_f64p              %.3f ms' % t(lambda: ops.stack_sigclip(frames, calib=calib, outputs=('moments_f64p',))))
print('moments_f64 legacy        %.3f ms' % t(lambda: ops.stack_sigclip(frames, calib=calib, outputs=('moments_f64',))))
for k in (1, 2, 4, 8):
    rows = parallel.stripe_rows(H, k)
    def stripes():
        for r0, r1 in rows:
            ops.stack_sigclip(frames[:, r0:r1], calib=parallel._slice_calib(calib, r0, r1), outputs=('moments_f64p',), moments_mean_only=True)
    print('%d stripes, one stream      %.3f ms' % (k, t(stripes)))
m = ops.stack_sigclip(frames, calib=calib, outputs=('moments_f64p',), moments_mean_only=True)['moments_f64p']
out = torch.empty((H, W), dtype=torch.float32, device=dev)
print('finalize f64p             %.3f ms' % t(lambda: ops.moments_finalize(dict(sum=m['sum'], count=m['count']), want_std=False, out_mean=out)))
import torch.distributed as dist
os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT='29533', RANK='0', WORLD_SIZE='1')
dist.init_process_group('nccl', device_id=dev)
for k in (1, 2, 4, 8):
    print('stack_nshard %d stripes     %.3f ms' % (k, t(lambda: parallel.stack_nshard(frames, calib, n_stripes=k, force_collective=True))))
print('all_reduce 268 MB (1 rank) %.3f ms' % t(lambda: dist.all_reduce(m['prefix'])))
dist.destroy_process_group()
