"""Multi-GPU stacking: one process per GPU, torch.distributed over RCCL/xGMI.

Two partitionings of a stack job:

* ``stack_nshard`` - the north-star layout: frames are sharded on the N axis, every rank reduces its
  own frames to per-pixel partial moments (sum, sum of squares, count of the locally clipped
  survivors), ONE all-reduce(sum) combines them and ``mean = sum / count``.  Semantics are
  *hierarchical* clipping (clip against the statistics of the rank's own frames, then combine):
  median-centred clipping is not decomposable over N shards, so for world_size > 1 the result is
  defined by "sigma_clipped_stats per shard, moments summed" - which is what the tests check - and
  equals the single-GPU result for world_size == 1.  The image is cut into row stripes; the
  all-reduce of stripe k runs on a communication stream while stripe k+1 is being reduced, so the
  collective hides behind the HBM-bound kernel instead of following it.
* ``stack_rowshard`` - exact for any centre/deviation function and needs no collective: every rank
  holds all N frames of a row block and reduces it on its own (use dist.all_gather afterwards if one
  rank wants the whole image).

The kernels are reached through ``astrophotography_amd.ops``; tests substitute CPU stand-ins for the
two device functions to run the collective plumbing under gloo.
"""
import torch
import torch.distributed as dist


def _world(group=None):
    if dist.is_available() and dist.is_initialized():
        return dist.get_world_size(group), dist.get_rank(group)
    return 1, 0


def stripe_rows(H, n_stripes):
    """Row ranges [(r0, r1), ...] of n_stripes nearly equal stripes (never empty)."""
    n_stripes = max(1, min(int(n_stripes), H))
    base, extra = divmod(H, n_stripes)
    out, r = [], 0
    for k in range(n_stripes):
        h = base + (1 if k < extra else 0)
        out.append((r, r + h))
        r += h
    return out


def shard_frames(n_total, world_size, rank):
    """Frame range of `rank` when n_total frames are dealt in contiguous blocks."""
    base, extra = divmod(n_total, world_size)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def _default_local_moments(frames, calib, r0, r1, clip):
    from . import ops
    sub = frames[:, r0:r1]
    c = None
    if calib is not None:
        c = dict(calib)
        for k in ('bias', 'dark', 'nflat'):
            if c.get(k) is not None:
                c[k] = c[k][r0:r1]
    return ops.stack_sigclip(sub, calib=c, outputs=('moments',), **clip)['moments']


def _default_finalize(moments, out_mean):
    from . import ops
    ops.moments_finalize(moments, want_std=False, out_mean=out_mean)


_COMM_STREAMS = {}


def _comm_stream(device, role='comm'):
    """Side streams per device (stripe all-reduces, alternating compute lanes), created once, not per call."""
    key = (device.type, device.index, role)
    if key not in _COMM_STREAMS:
        _COMM_STREAMS[key] = torch.cuda.Stream(device=device)
    return _COMM_STREAMS[key]


def stack_nshard(frames_local, calib=None, sigma=3.0, maxiters=5, cenfunc='median', stdfunc='std',
                 n_stripes=8, group=None, local_moments=None, finalize=None, return_moments=False, force_collective=False,
                 exchange_sumsq=None):
    """N-sharded clipped mean: frames_local[n_local, H, W] on this rank -> mean[H, W] on every rank.

    One all-reduce per stripe on a side stream, overlapped with the reduction of the next stripe.  The
    moments are laid out (sum, count, sum of squares); the mean needs only the first two, so unless
    `exchange_sumsq` (default: `return_moments`) asks for the full set the collective carries the contiguous
    [2, stripe, W] prefix - 8 instead of 12 bytes per pixel over xGMI.
    """
    if exchange_sumsq is None:
        exchange_sumsq = return_moments
    local_moments = local_moments or _default_local_moments
    finalize = finalize or _default_finalize
    world, _ = _world(group)
    n_local, H, W = frames_local.shape
    clip = dict(sigma=sigma, maxiters=maxiters, cenfunc=cenfunc, stdfunc=stdfunc)
    on_gpu = frames_local.is_cuda
    collective = (world > 1) or (force_collective and dist.is_available() and dist.is_initialized())
    stripes = stripe_rows(H, n_stripes if collective else 1)
    parts = []
    mean = torch.empty((H, W), dtype=torch.float32, device=frames_local.device)
    if collective and on_gpu:
        dev = frames_local.device
        comm = _comm_stream(dev)
        main = torch.cuda.current_stream()
        # two alternating compute streams: stripe k + 1 starts filling the CUs that stripe k's last workgroups
        # leave idle (kernels on ONE stream serialise, and every stripe has a drain phase)
        lanes = [_comm_stream(dev, 'compute0'), _comm_stream(dev, 'compute1')]
        for st in lanes + [comm]:
            st.wait_stream(main)
        for k, (r0, r1) in enumerate(stripes):
            cs = lanes[k % 2]
            with torch.cuda.stream(cs):
                m = local_moments(frames_local, calib, r0, r1, clip)
                ev = torch.cuda.Event()
                ev.record(cs)
                m.record_stream(cs)
            with torch.cuda.stream(comm):
                # exchange and finalise stripe k on the side stream while the compute streams reduce the next stripes
                comm.wait_event(ev)
                dist.all_reduce(m if exchange_sumsq else m[:2], op=dist.ReduceOp.SUM, group=group)
                finalize(m, mean[r0:r1])
                m.record_stream(comm)
            parts.append(m)
        for st in lanes + [comm]:
            main.wait_stream(st)
    else:
        for (r0, r1) in stripes:
            m = local_moments(frames_local, calib, r0, r1, clip)
            if collective:
                dist.all_reduce(m if exchange_sumsq else m[:2], op=dist.ReduceOp.SUM, group=group)
            finalize(m, mean[r0:r1])
            parts.append(m)
    if return_moments:
        return mean, (parts[0] if len(parts) == 1 else torch.cat(parts, dim=1))
    return mean


def stack_rowshard(frames_rows, calib=None, sigma=3.0, maxiters=5, cenfunc='median', stdfunc='std', outputs=('mean',)):
    """Exact path: this rank's row block frames_rows[N, h, W] (all N frames) -> its block of the result."""
    from . import ops
    return ops.stack_sigclip(frames_rows, sigma=sigma, maxiters=maxiters, cenfunc=cenfunc, stdfunc=stdfunc,
                             calib=calib, outputs=outputs)
