"""Multi-GPU stacking: one process per GPU, torch.distributed over RCCL/xGMI.

Two partitionings of a stack job:

* ``stack_nshard`` - the north-star layout: frames are sharded on the N axis, every rank reduces its
  own frames to per-pixel partial moments of the locally clipped survivors, ONE all-reduce(sum) per row
  stripe combines them and ``mean = sum / count``.  Semantics are *hierarchical* clipping (clip against
  the statistics of the rank's own frames, then combine): median-centred clipping is not decomposable
  over N shards, so for world_size > 1 the result is defined by "sigma_clipped_stats per shard,
  moments summed" - which is what the tests check - and equals the single-GPU result for
  world_size == 1.  The image is cut into row stripes; the all-reduce of stripe k runs on a
  communication stream while stripe k+1 is being reduced, so the collective hides behind the kernel
  instead of following it.

  Exchange forms (``exchange=``):
    'rs'  (default for world > 1 when a stripe's rows divide by the world size; round 4, slimmed in round 5)  the moment planes
                     are REDUCE-SCATTERED by rows - every rank receives the combined (sum, count[, sumsq]) of its own
                     1/world of the stripe's rows, finalises those rows, and the float32 mean (and std) rows are
                     ALL-GATHERED.  Round 5: the count no longer rides as a float64 plane - the kernels write the float64
                     sum + INT32 count layout (include/apgpu.h, moments_f64 = 1) and the count is exchanged as its own
                     4-byte plane, or as a 2-byte float16 plane (`count_dtype=torch.float16`: exact while the whole job has at
                     most 2048 frames - integers up to 2048 are float16 numbers and so are all their partial sums): per pixel
                     (world-1)/world x (8 + 4 + 4) = 14 bytes leave a rank at 8 ranks (12.25 with the float16 count) where
                     round 4's float64 count made it 17.5 and the all-reduce form sends 28; a rank finalises 1/world of the
                     pixels.  Same float64 combine rounded once (the ranks' sums are added in float64 either way; the
                     order of the additions is the collective's).  Stripes whose rows do not divide fall back to 'f64'.
    'f64'            ONE all-reduce per stripe (the north star's literal form) of the packed float64 moment planes (sum, count[, sumsq]) of include/apgpu.h's layout 3 - 16 bytes per
                     pixel, 24 with a std: the count rides along as a float64 (exact to 2^53), so a single call carries
                     everything; the ranks' float64 partial sums are added in float64 and the combined mean is rounded
                     once to float32.  (A mean-only exchange lets the moment kernel take its float32 fast path - a rank's
                     sum is then n c + the float32 sum of the deviations, within ~1e-8 of the float64 sum of its
                     survivors; exact=True forces the float64 clip, and then the result IS the float64 combine of
                     SURVEY 8(e) rounded once);
    'f32'            float32 sum + float32 count (8 bytes): every rank rounds its sum to float32 and
                     RCCL adds in float32 - about 1e-7 relative per rank, mean only (no std: float32
                     sums of squares about zero cancel catastrophically for CCD-range data).
  ``hier_chunk``: a rank clips its own frames in chunks of that many (ops.stack_sigclip_chunked, moments accumulated on
  the device) - with chunk = n_total / 8 a job has the SAME semantics on 1, 2, 4 and 8 ranks ("8 shards of n_total / 8
  frames, clipped per shard, moments added"), which is what makes a strong-scaling curve compare like with like.

* ``stack_rowshard`` - exact for any centre/deviation function and needs no data-path collective:
  every rank holds all N frames of its row block (``row_block``) and reduces it on its own;
  ``gather_rows`` assembles the image on every rank when one is wanted.

The kernels are reached through ``astrophotography_amd.ops``; the CPU tests substitute stand-ins for
the two device functions to run the collective plumbing under gloo.
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


def row_block(H, world_size, rank):
    """Row range of `rank` in the row-sharded layout (contiguous blocks, the first H % world get one more)."""
    return shard_frames(H, world_size, rank)


def _slice_calib(calib, r0, r1):
    if calib is None:
        return None
    c = dict(calib)
    for k in ('bias', 'dark', 'nflat'):
        if c.get(k) is not None:
            c[k] = c[k][r0:r1]
    return c


def _default_local_moments(frames, calib, r0, r1, clip, exchange, want_std=False, hier_chunk=None, exact=False):
    """Partial moments of rows [r0, r1): dict(sum, count[, sumsq], prefix[, buffer]) of [r1 - r0, W] planes.
    exact: APGPU_STACK_EXACT_MOMENTS - the float64 clip only, so that the float64 sums ARE the float64 sums of the survivors
    (by default a mean-only exchange lets the kernel take its float32 fast path: sum = n c + float32 S)."""
    from . import ops
    sub = frames[:, r0:r1]
    c = _slice_calib(calib, r0, r1)
    chunked = hier_chunk is not None and sub.shape[0] > hier_chunk
    if exchange == 'f32':
        if chunked:
            raise ValueError("hierarchical chunks accumulate float64 moments: use exchange='f64'")
        m = ops.stack_sigclip(sub, calib=c, outputs=('moments',), exact=exact, **clip)['moments']
        return dict(sum=m[0], count=m[1], prefix=m[:2])         # (sum, count) is one contiguous float32 block
    packed = exchange != 'f64i'                              # 'f64': packed float64 planes (one all-reduce); 'f64i': float64 sums + int32 count ('rs')
    if chunked:
        return ops.stack_sigclip_chunked(sub, chunk=hier_chunk, want_std=want_std, packed=packed, finalize=False, calib=c, exact=exact, **clip)
    key = 'moments_f64p' if packed else 'moments_f64'
    return ops.stack_sigclip(sub, calib=c, outputs=(key,), moments_mean_only=not want_std, exact=exact, **clip)[key]


def _default_finalize(m, out_mean, out_std, exchange):
    from . import ops
    if exchange == 'f32':
        ops.moments_finalize(m['prefix'], want_std=False, out_mean=out_mean)
        return
    if out_std is None:
        ops.moments_finalize(dict(sum=m['sum'], count=m['count']), want_std=False, out_mean=out_mean)
    else:
        _, std = ops.moments_finalize(dict(sum=m['sum'], sumsq=m['sumsq'], count=m['count']), want_std=True, out_mean=out_mean)
        out_std.copy_(std)


def _exchange(m, exchange, want_std, group):
    """THE all-reduce of one stripe: the contiguous (sum, count) planes - float32 or float64 - or, with a std, the whole
    packed float64 buffer (sum, count, sumsq)."""
    if exchange == 'f64i':
        # (the float64 sum + int32 count layout of 'rs' has no single-dtype block: a stripe that cannot be reduce-scattered
        # - rows not divisible by the world size - all-reduces its planes one by one)
        for k in (('sum', 'count', 'sumsq') if want_std else ('sum', 'count')):
            dist.all_reduce(m[k], op=dist.ReduceOp.SUM, group=group)
        return
    dist.all_reduce(m['buffer'] if (want_std and exchange == 'f64') else m['prefix'], op=dist.ReduceOp.SUM, group=group)


def _exchange_rs(m, want_std, group, finalize, out_mean, out_std, count_dtype=torch.int32, gather=True):
    """The reduce-scatter form of one stripe (rows divisible by the world size): the float64 sum plane [h, W] (and sumsq,
    with a std) and the count plane - int32 as the kernels write it, or narrowed to float16 - are reduce-scattered by rows,
    the rank finalises ITS rows, the float32 result rows are all-gathered into out_mean (out_std).  Returns the rank's
    combined moment rows (dict(sum, count (int32)[, sumsq], rows=(a, b)))."""
    world, rank = _world(group)
    h, W = m['sum'].shape
    hb = h // world
    own = {}
    for k in (('sum', 'sumsq') if want_std else ('sum',)):
        plane = m[k]
        own[k] = torch.empty((hb, W), dtype=plane.dtype, device=plane.device)
        dist.reduce_scatter_tensor(own[k], plane, op=dist.ReduceOp.SUM, group=group)
    cnt = m['count']
    if cnt.dtype != count_dtype:
        cnt = cnt.to(count_dtype)                            # (the packed layout's float64 plane, or int32 -> float16)
    got = torch.empty((hb, W), dtype=count_dtype, device=cnt.device)
    dist.reduce_scatter_tensor(got, cnt.contiguous(), op=dist.ReduceOp.SUM, group=group)
    own['count'] = got if got.dtype == torch.int32 else got.to(torch.int32)
    if not gather:
        # the result stays ROW-DISTRIBUTED: the rank finalises its rows straight into the output image, nothing is gathered
        # (what a job needs whose next step is row-parallel too - or whose product every rank writes as its own strip)
        finalize(own, out_mean[rank * hb:(rank + 1) * hb], out_std[rank * hb:(rank + 1) * hb] if want_std else None, 'f64i')
        own['rows'] = (rank * hb, (rank + 1) * hb)
        return own
    my_mean = torch.empty((hb, W), dtype=torch.float32, device=out_mean.device)
    my_std = torch.empty((hb, W), dtype=torch.float32, device=out_mean.device) if want_std else None
    finalize(own, my_mean, my_std, 'f64i')
    dist.all_gather_into_tensor(out_mean, my_mean, group=group)
    if want_std:
        dist.all_gather_into_tensor(out_std, my_std, group=group)
    own['rows'] = (rank * hb, (rank + 1) * hb)
    own['_keep'] = (my_mean, my_std)
    return own


def exchange_bytes_per_pixel(exchange='f64', want_std=False, count_bytes=4):
    """Bytes per output pixel of the moment planes each rank contributes to the exchange ('rs' / 'f64i': float64 sum[, sumsq] +
    a count plane of count_bytes)."""
    if exchange == 'f32':
        return 8
    if exchange in ('rs', 'f64i'):
        return (16 if want_std else 8) + count_bytes
    return 24 if want_std else 16


def exchange_bytes_on_wire(exchange, world, n_pixels, want_std=False, count_bytes=4, gather=True):
    """Bytes one rank SENDS per step for n_pixels output pixels (ring collectives): all-reduce = 2 (w-1)/w x payload;
    'rs' = (w-1)/w x (payload + the float32 result planes that are all-gathered - nothing with gather=False)."""
    if world <= 1:
        return 0
    f = (world - 1) / world
    payload = exchange_bytes_per_pixel(exchange, want_std, count_bytes)
    if exchange == 'rs':
        return int(f * (payload + ((8 if want_std else 4) if gather else 0)) * n_pixels)
    return int(2 * f * payload * n_pixels)


def default_stripes(H, W, exchange='f64', want_std=False):
    """Row stripes per step: enough to overlap the exchange with the reduction, few enough that every all-reduce stays a
    large transfer (>= 64 MB) - 4 for a 4096 x 4096 image, never more than 8."""
    payload = H * W * exchange_bytes_per_pixel(exchange, want_std)
    if exchange in ('rs', 'f64i'):
        # a power of two (rows stay divisible by the world size, or the stripe falls back to all-reduces), >= 48 MB per stripe
        n = int(max(1, min(8, payload // (48 << 20))))
        return 1 << (n.bit_length() - 1)
    return int(max(1, min(8, payload // (64 << 20))))


_COMM_STREAMS = {}


def _comm_stream(device, role='comm'):
    """Side streams per device (stripe all-reduces, alternating compute lanes), created once, not per call."""
    key = (device.type, device.index, role)
    if key not in _COMM_STREAMS:
        _COMM_STREAMS[key] = torch.cuda.Stream(device=device)
    return _COMM_STREAMS[key]


def _record(m, stream):
    for t in m.values():
        if torch.is_tensor(t):
            t.record_stream(stream)


def stack_nshard(frames_local, calib=None, sigma=3.0, maxiters=5, cenfunc='median', stdfunc='std',
                 n_stripes=None, group=None, local_moments=None, finalize=None, return_moments=False, force_collective=False,
                 exchange=None, want_std=False, hier_chunk=None, timings=None, exact=False, count_dtype=torch.int32, gather=True):
    """N-sharded clipped mean: frames_local[n_local, H, W] on this rank -> mean[H, W] on every rank
    (want_std: (mean, std)).

    One exchange per stripe on a side stream, overlapped with the reduction of the next stripes.  `exchange`
    selects the payload (module docstring); want_std needs 'f64'.  return_moments appends the combined per-stripe
    moment dicts (sum, count[, sumsq]) for inspection.  n_stripes None = default_stripes().  hier_chunk: see the module
    docstring.  timings: a list that receives one (start, end) pair of CUDA events per stripe around its all-reduce on the
    communication stream (bench.py's exchange_ms).  count_dtype ('rs'): torch.int32, or torch.float16 when the WHOLE job has
    at most 2048 frames (the caller's knowledge: a rank only sees its own).
    gather=False ('rs' only, every stripe's rows divisible by the world size): no all-gather - the returned image holds THIS
    rank's rows of every stripe (the rows own_rows(H, n_stripes, world, rank) names) and is undefined elsewhere: 10 bytes per
    pixel x (world - 1) / world leave a rank per step (float16 count) where the gathered form sends 14 and the all-reduce 28.
    """
    if exchange is None:
        exchange = 'rs'
    if exchange not in ('rs', 'f64', 'f32'):
        raise ValueError("exchange must be 'rs', 'f64' or 'f32'")
    if want_std and exchange == 'f32':
        raise ValueError("a standard deviation needs exchange='rs' or 'f64' (float32 sums of squares cancel)")
    if local_moments is None:
        import functools
        local_moments = functools.partial(_default_local_moments, want_std=want_std, hier_chunk=hier_chunk, exact=exact)
    finalize = finalize or _default_finalize
    world, _ = _world(group)
    n_local, H, W = frames_local.shape
    if count_dtype not in (torch.int32, torch.float16):
        raise ValueError('count_dtype must be torch.int32 or torch.float16')
    payload = 'f64i' if exchange == 'rs' else exchange       # the moment layout the kernels write
    if n_stripes is None:
        n_stripes = default_stripes(H, W, payload, want_std)
    clip = dict(sigma=sigma, maxiters=maxiters, cenfunc=cenfunc, stdfunc=stdfunc)
    if not gather:
        if exchange != 'rs':
            raise ValueError("gather=False belongs to exchange='rs'")
        if world > 1 and any((b - a) % world for a, b in stripe_rows(H, n_stripes)):
            raise ValueError('gather=False needs every stripe\'s rows divisible by the world size (%d rows in %d stripes, world %d)' % (H, n_stripes, world))
    on_gpu = frames_local.is_cuda
    collective = (world > 1) or (force_collective and dist.is_available() and dist.is_initialized())
    stripes = stripe_rows(H, n_stripes if collective else 1)
    parts = []
    mean = torch.empty((H, W), dtype=torch.float32, device=frames_local.device)
    std = torch.empty((H, W), dtype=torch.float32, device=frames_local.device) if want_std else None
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
                m = local_moments(frames_local, calib, r0, r1, clip, payload)
                ev = torch.cuda.Event()
                ev.record(cs)
                _record(m, cs)
            with torch.cuda.stream(comm):
                # exchange and finalise stripe k on the side stream while the compute streams reduce the next stripes
                comm.wait_event(ev)
                if timings is not None:
                    t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    t0.record(comm)
                if exchange == 'rs' and (r1 - r0) % world == 0:
                    own = _exchange_rs(m, want_std, group, finalize, mean[r0:r1], std[r0:r1] if want_std else None, count_dtype, gather)
                    _record(own, comm)
                    if timings is not None:
                        t1.record(comm)
                        timings.append((t0, t1))
                    _record(m, comm)
                    m = own
                else:
                    _exchange(m, payload, want_std, group)
                    if timings is not None:
                        t1.record(comm)
                        timings.append((t0, t1))
                    finalize(m, mean[r0:r1], std[r0:r1] if want_std else None, payload)
                    _record(m, comm)
            parts.append(m)
        for st in lanes + [comm]:
            main.wait_stream(st)
    else:
        for (r0, r1) in stripes:
            m = local_moments(frames_local, calib, r0, r1, clip, payload)
            if collective and exchange == 'rs' and (r1 - r0) % world == 0:
                m = _exchange_rs(m, want_std, group, finalize, mean[r0:r1], std[r0:r1] if want_std else None, count_dtype, gather)
            else:
                if collective:
                    _exchange(m, payload, want_std, group)
                finalize(m, mean[r0:r1], std[r0:r1] if want_std else None, payload)
            parts.append(m)
    out = (mean, std) if want_std else mean
    if return_moments:
        return out, parts
    return out


def own_rows(H, n_stripes, world_size, rank):
    """The global row ranges [(a, b), ...] a rank holds after stack_nshard(.., exchange='rs', gather=False): its 1 / world of
    every stripe."""
    out = []
    for r0, r1 in stripe_rows(H, n_stripes):
        hb = (r1 - r0) // world_size
        out.append((r0 + rank * hb, r0 + (rank + 1) * hb))
    return out


def stack_rowshard(frames_rows, calib=None, sigma=3.0, maxiters=5, cenfunc='median', stdfunc='std', outputs=('mean',),
                   method='sigclip'):
    """Exact path: this rank's row block frames_rows[N, h, W] (all N frames; masters in `calib` cut to the same
    rows) -> its block of the result.  No collective: every output pixel depends only on its own column."""
    from . import ops
    if method == 'median':
        return dict(median=ops.stack_median(frames_rows, calib=calib))
    return ops.stack_sigclip(frames_rows, sigma=sigma, maxiters=maxiters, cenfunc=cenfunc, stdfunc=stdfunc,
                             calib=calib, outputs=outputs)


def gather_rows(block, H, group=None):
    """All-gather of the ranks' row blocks block[h_r, W] (row_block layout) into the full [H, W] image on every rank.
    This is output assembly (4 bytes per pixel, once), not part of the reduction."""
    world, rank = _world(group)
    if world == 1:
        return block
    W = block.shape[1]
    lo, hi = row_block(H, world, rank)
    if block.shape[0] != hi - lo:
        raise ValueError('rank %d holds %d rows, the row_block layout expects %d' % (rank, block.shape[0], hi - lo))
    full = torch.empty((H, W), dtype=block.dtype, device=block.device)
    if H % world == 0:
        dist.all_gather_into_tensor(full, block.contiguous(), group=group)
        return full
    hmax = -(-H // world)                                   # ragged blocks: pad to the tallest, gather, cut
    padded = torch.zeros((hmax, W), dtype=block.dtype, device=block.device)
    padded[:hi - lo] = block
    allb = torch.empty((world * hmax, W), dtype=block.dtype, device=block.device)
    dist.all_gather_into_tensor(allb, padded, group=group)
    for r in range(world):
        a, b = row_block(H, world, r)
        full[a:b] = allb[r * hmax:r * hmax + (b - a)]
    return full
