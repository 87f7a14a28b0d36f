"""Array-level entry points: torch tensors in HBM -> libapgpu.so kernels (include/apgpu.h).

These are the slab/array forms underneath the file-based Ap* classes (the reference has no in-memory
API for this path: "TODO: Allow calibration of images in memory", core/ApCalibrate.py:3).  PyTorch
only owns device memory and streams here; every computation is a hand-written HIP kernel.  All calls are
asynchronous on the current torch stream.
"""
import ctypes as C

import numpy as np
import torch

from . import _lib
from ._lib import APGPU_F32, APGPU_F64, APGPU_U16, StackArgs, check


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _ptr(t):
    return None if t is None else C.c_void_p(t.data_ptr())


def _aligned16(t):
    """A contiguous tensor whose storage starts on a 16-byte boundary (the vectorised kernels require it): a frame cut out
    of a slab with an odd pixel count is copied once."""
    t = t.contiguous()
    return t if t.data_ptr() % 16 == 0 else t.clone()


def _need_cuda(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise ValueError('libapgpu operates on device tensors (got a %s tensor); there is no CPU path' % t.device)


def _f32c(t, name):
    if t is None:
        return None
    if t.dtype != torch.float32:
        raise TypeError('%s must be float32, got %s' % (name, t.dtype))
    return t.contiguous()


def _raw_dtype(t):
    if t.dtype == torch.float32:
        return APGPU_F32
    if t.dtype in (torch.uint16, torch.int16):          # int16 storage is reinterpreted as uint16
        return APGPU_U16
    raise TypeError('frame data must be float32 or uint16, got %s' % t.dtype)


def to_device_u16(a, device='cuda'):
    """numpy uint16 array -> device tensor with dtype torch.uint16."""
    a = np.ascontiguousarray(a, dtype=np.uint16)
    return torch.from_numpy(a.view(np.int16)).to(device).view(torch.uint16)


def _per_frame(x, n, device):
    """python float / sequence / tensor -> float32 device tensor [n] (float32 cast like numpy's weak scalar)."""
    if x is None:
        return None
    if torch.is_tensor(x):
        t = x.to(device=device, dtype=torch.float32).reshape(-1)
        if t.numel() == 1 and n > 1:
            t = t.expand(n)
        return t.contiguous()
    a = np.broadcast_to(np.asarray(x, dtype=np.float64), (n,)).astype(np.float32)
    return torch.from_numpy(np.ascontiguousarray(a)).to(device)


# ---------------------------------------------------------------------------------------------------
def _dtype_tag(t, what, allow_u16=False):
    if t.dtype == torch.float32:
        return APGPU_F32
    if t.dtype == torch.float64:
        return APGPU_F64
    if allow_u16 and t.dtype in (torch.uint16, torch.int16):
        return APGPU_U16
    raise TypeError('%s must be float32 or float64%s, got %s' % (what, ' or uint16' if allow_u16 else '', t.dtype))


def flat_normalize(flat):
    """A1 ApCalibrate._generate_flat (ApCalibrate.py:166-190): returns (nflat, norm[1] device tensor), both in the
    flat's own dtype - float32, or float64 for a float64 master (the reference keeps float FITS data as stored,
    ApCalibrate.py:301-305)."""
    _need_cuda(flat)
    lib = _lib.load()
    if flat.dtype == torch.float64:
        flat = flat.contiguous()
        n = flat.numel()
        ws_bytes = lib.apgpu_flat_normalize_f64_ws_bytes(n)
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=flat.device)
        nflat = torch.empty_like(flat)
        norm = torch.empty(1, dtype=torch.float64, device=flat.device)
        check(lib.apgpu_flat_normalize_f64(_ptr(flat), _ptr(nflat), _ptr(norm), n, _ptr(ws), ws_bytes, _stream()))
        return nflat, norm
    flat = _f32c(flat, 'flat')
    n = flat.numel()
    ws_bytes = lib.apgpu_flat_normalize_ws_bytes(n)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=flat.device)
    nflat = torch.empty_like(flat)
    norm = torch.empty(1, dtype=torch.float32, device=flat.device)
    check(lib.apgpu_flat_normalize_f32(_ptr(flat), _ptr(nflat), _ptr(norm), n, _ptr(ws), ws_bytes, _stream()))
    return nflat, norm


def calibrate(raw, bias, dark, nflat, exp_ratio, pedestal=None, dark_still_biased=False, out=None):
    """A2 ApCalibrate.calibrate arithmetic (ApCalibrate.py:439-464) on raw[H,W] or a slab raw[N,H,W]."""
    _need_cuda(raw, bias, dark, nflat)
    lib = _lib.load()
    raw = raw.contiguous()
    if any(t is not None and t.dtype == torch.float64 for t in (raw, bias, dark, nflat)):
        return _calibrate_mixed(raw, bias, dark, nflat, exp_ratio, pedestal, dark_still_biased, out)
    dt = _raw_dtype(raw)
    single = raw.dim() == 2
    N = 1 if single else raw.shape[0]
    P = raw[0].numel() if not single else raw.numel()
    bias, dark, nflat = _f32c(bias, 'bias'), _f32c(dark, 'dark'), _f32c(nflat, 'nflat')
    for nm, t in (('bias', bias), ('dark', dark), ('nflat', nflat)):
        if t is not None and t.numel() != P:
            raise RuntimeError('%s has %d pixels, frames have %d' % (nm, t.numel(), P))
    e = _per_frame(exp_ratio, N, raw.device)
    ped = _per_frame(pedestal, N, raw.device)
    if out is None:
        out = torch.empty(raw.shape, dtype=torch.float32, device=raw.device)
    check(lib.apgpu_calibrate(_ptr(raw), dt, _ptr(bias), _ptr(dark), _ptr(nflat), _ptr(e), _ptr(ped),
                              int(bool(dark_still_biased)), _ptr(out), N, P, _stream()))
    return out


def _per_frame_f64(x, n, device):
    if x is None:
        return None
    if torch.is_tensor(x):
        t = x.to(device=device, dtype=torch.float64).reshape(-1)
        return (t.expand(n) if t.numel() == 1 and n > 1 else t).contiguous()
    return torch.from_numpy(np.array(np.broadcast_to(np.asarray(x, dtype=np.float64), (n,)))).to(device)


def _calibrate_mixed(raw, bias, dark, nflat, exp_ratio, pedestal, dark_still_biased, out):
    """A2 with float64 inputs (apgpu_calibrate_mixed): NumPy's per-operation promotion; the result is float64 if any
    participating array is float64."""
    lib = _lib.load()
    single = raw.dim() == 2
    N = 1 if single else raw.shape[0]
    P = raw.numel() // N
    rt = _dtype_tag(raw, 'raw', allow_u16=True)
    bias, dark = bias.contiguous(), dark.contiguous()
    nflat = None if nflat is None else nflat.contiguous()
    bt, dk = _dtype_tag(bias, 'bias'), _dtype_tag(dark, 'dark')
    nt = _dtype_tag(nflat, 'nflat') if nflat is not None else APGPU_F32
    for nm, t in (('bias', bias), ('dark', dark), ('nflat', nflat)):
        if t is not None and t.numel() != P:
            raise RuntimeError('%s has %d pixels, frames have %d' % (nm, t.numel(), P))
    r64, b64, d64 = rt == APGPU_F64, bt == APGPU_F64, dk == APGPU_F64
    t3 = (r64 or b64) or ((d64 or b64) if dark_still_biased else d64)
    t4 = (t3 or nt == APGPU_F64) if nflat is not None else t3
    odt = torch.float64 if t4 else torch.float32
    e = _per_frame_f64(exp_ratio, N, raw.device)
    ped = _per_frame_f64(pedestal, N, raw.device)
    if out is None:
        out = torch.empty(raw.shape, dtype=odt, device=raw.device)
    elif out.dtype != odt or out.shape != raw.shape or not out.is_contiguous():
        raise TypeError('out must be a contiguous %s tensor of the raw shape' % odt)
    check(lib.apgpu_calibrate_mixed(_ptr(raw), rt, _ptr(bias), bt, _ptr(dark), dk, _ptr(nflat), nt, _ptr(e), _ptr(ped),
                                    int(bool(dark_still_biased)), _ptr(out), APGPU_F64 if t4 else APGPU_F32, N, P, _stream()))
    return out


def _stack_args(frames, calib, pixmask, keep):
    if frames.dim() < 2:
        raise ValueError('frames must be [N, ...]')
    N = frames.shape[0]
    shp = tuple(frames.shape[1:])
    P = frames[0].numel()
    # a row stripe frames_full[:, r0:r1] of a contiguous slab is reduced in place (frame_stride > P)
    if not (frames[0].is_contiguous() and (N == 1 or frames.stride(0) >= P)):
        frames = frames.contiguous()
    a = StackArgs()
    a.frames = frames.data_ptr()
    a.dtype = _raw_dtype(frames)
    a.n_frames = N
    a.n_pixels = P
    a.frame_stride = frames.stride(0) if N > 1 else P
    keep.append(frames)
    if calib is not None:
        bias, dark = _f32c(calib['bias'], 'bias'), _f32c(calib['dark'], 'dark')
        nflat = _f32c(calib.get('nflat'), 'nflat')
        e = _per_frame(calib['exp_ratio'], N, frames.device)
        ped = _per_frame(calib.get('pedestal'), N, frames.device)
        for nm, t in (('bias', bias), ('dark', dark), ('nflat', nflat)):
            if t is not None and t.numel() != P:
                raise RuntimeError('%s has %d pixels, frames have %d' % (nm, t.numel(), P))
        keep += [bias, dark, nflat, e, ped]
        a.bias, a.dark = bias.data_ptr(), dark.data_ptr()
        a.nflat = nflat.data_ptr() if nflat is not None else None
        a.exp_ratio = e.data_ptr()
        a.pedestal = ped.data_ptr() if ped is not None else None
        a.dark_still_biased = int(bool(calib.get('dark_still_biased', False)))
    if pixmask is not None:
        if pixmask.dtype != torch.uint8 or pixmask.numel() != P:
            raise TypeError('pixmask must be uint8 with one entry per pixel')
        pixmask = pixmask.contiguous()
        keep.append(pixmask)
        a.pixmask = pixmask.data_ptr()
    return a, N, P, shp, frames.device


# Workspaces of the stack's two-kernel scheme (apgpu_stack_args.workspace): one per (device, stream, pixel count), zeroed once
# - the library leaves the part that must be zero as it found it.  Keyed by the stream because a workspace must not be shared
# by calls that may run concurrently (parallel.stack_nshard alternates two compute streams).
_stack_ws = {}


def stack_workspace(n_pixels, device):
    lib = _lib.load()
    dev = torch.device(device)
    key = (dev.index if dev.index is not None else torch.cuda.current_device(), torch.cuda.current_stream(dev).cuda_stream, int(n_pixels))
    ws = _stack_ws.get(key)
    if ws is None:
        zero = C.c_size_t(0)
        nbytes = lib.apgpu_stack_ws_bytes(int(n_pixels), C.byref(zero))
        ws = torch.empty((nbytes + 7) // 8, dtype=torch.int64, device=dev)
        ws[:(zero.value + 7) // 8].zero_()
        if len(_stack_ws) >= 32:                             # many image sizes in one process: forget the oldest
            _stack_ws.pop(next(iter(_stack_ws)))
        _stack_ws[key] = ws
    return ws


def _attach_workspace(a, P, dev):
    ws = stack_workspace(P, dev)
    a.workspace = ws.data_ptr()
    a.workspace_bytes = ws.numel() * 8
    return ws


def stack_redo_stats(reset=False):
    """What the fast kernels of the stack left to the redo pass, summed over this process's workspaces: dict(calls, pixels,
    pixels_listed, blocks_given_up (64-pixel blocks), fraction = share of the pixels that did not finish on the fast path).  Synchronises."""
    tot = np.zeros(4, dtype=np.int64)
    for ws in _stack_ws.values():
        torch.cuda.synchronize(ws.device)
        o = _lib.STACK_WS_STATS_OFFSET // 8
        tot += ws[o:o + 4].cpu().numpy()
        if reset:
            ws[o:o + 4].zero_()
    calls, pixels, listed, blocks = (int(x) for x in tot)
    return dict(calls=calls, pixels=pixels, pixels_listed=listed, blocks_given_up=blocks,
                fraction=((listed + 64 * blocks) / pixels if pixels else 0.0))


def stack_sigclip(frames, sigma=3.0, sigma_lower=None, sigma_upper=None, maxiters=5, cenfunc='median',
                  stdfunc='std', calib=None, pixmask=None, outputs=('mean',), exact=False, moments_mean_only=False,
                  single_kernel=False, workspace=True, nonfinite_unclipped=False):
    """Per-pixel sigma-clipped reduction along N = astropy sigma_clipped_stats(cube, axis=0)
    (sigma_clipping.py:298-383, 924-937), optionally fused with the calibration of each value.

    calib: None or dict(bias, dark, nflat=None, exp_ratio, pedestal=None, dark_still_biased=False).
    outputs: any of 'mean', 'median', 'std', 'count' (float32 / int32 planes), 'mean_f64', 'std_f64' (the unrounded
    float64 statistics, as ccdproc.combine keeps them), 'moments' (float32 [3, ...]: sum, count, sumsq) or
    'moments_f64' (dict(sum, sumsq: float64 planes, count: int32 plane) - views of one 20-byte-per-pixel buffer) or
    'moments_f64p' (the packed float64 layout float64 [3, ...] = sum, count, sumsq: dict(sum, count, sumsq, buffer,
    prefix = buffer[:2]) - what parallel.stack_nshard all-reduces in ONE call per stripe)
    -> dict of device tensors.
    exact: APGPU_STACK_EXACT_MOMENTS (float64 clip only: the mean is the float64 mean of the survivors rounded once);
    moments_mean_only: APGPU_STACK_MOMENTS_MEAN (the float64 moments will only be turned into a mean).
    single_kernel: APGPU_STACK_SINGLE_KERNEL (one complete kernel instead of the fast kernel + redo pass pair);
    nonfinite_unclipped: APGPU_STACK_NONFINITE_UNCLIPPED (stdfunc='mad_std' only): a column holding a non-finite value is not
    clipped - ccdproc >= 2.2's Combiner.sigma_clipping through astropy.stats.sigma_clip (golden group G12, arrays c*_b_*);
    workspace: True = this module's cached workspace for the two-kernel scheme (stack_workspace), False = none (the
    library then allocates a stream-ordered temporary per call), or a tensor from stack_workspace().
    """
    _need_cuda(frames)
    lib = _lib.load()
    keep = []
    a, N, P, shp, dev = _stack_args(frames, calib, pixmask, keep)
    a.center = _lib.CENTER[cenfunc]
    a.dev = _lib.DEV[stdfunc]
    a.maxiters = -1 if maxiters is None else int(maxiters)
    a.sigma_lower = float(sigma if sigma_lower is None else sigma_lower)
    a.sigma_upper = float(sigma if sigma_upper is None else sigma_upper)
    res = {}
    for k in outputs:
        if k in ('mean', 'median', 'std'):
            res[k] = torch.empty(shp, dtype=torch.float32, device=dev)
        elif k == 'count':
            res[k] = torch.empty(shp, dtype=torch.int32, device=dev)
        elif k in ('mean_f64', 'std_f64'):
            res[k] = torch.empty(shp, dtype=torch.float64, device=dev)
        elif k == 'moments':
            if a.moments:
                raise ValueError('ask for one moment layout')
            res[k] = torch.empty((3,) + shp, dtype=torch.float32, device=dev)
            a.moments = res[k].data_ptr()
            continue
        elif k in ('moments_f64', 'moments_f64p'):
            if a.moments:
                raise ValueError('ask for one moment layout')
            res[k] = alloc_moments_f64(shp, dev, packed=(k == 'moments_f64p'))
            a.moments = res[k]['buffer'].data_ptr()
            a.moments_f64 = 3 if k == 'moments_f64p' else 1
            continue
        else:
            raise ValueError('unknown output %r' % (k,))
        setattr(a, k, res[k].data_ptr())
    a.flags = ((_lib.STACK_EXACT_MOMENTS if exact else 0) | (_lib.STACK_MOMENTS_MEAN if moments_mean_only else 0) |
               (_lib.STACK_SINGLE_KERNEL if single_kernel else 0) | (_lib.STACK_NONFINITE_UNCLIPPED if nonfinite_unclipped else 0))
    if workspace is True:
        keep.append(_attach_workspace(a, P, dev))
    elif workspace is not False and workspace is not None:
        a.workspace, a.workspace_bytes = workspace.data_ptr(), workspace.numel() * workspace.element_size()
    check(lib.apgpu_stack_sigclip(C.byref(a), _stream()))
    return res


def stack_sigclip_chunked(frames, chunk=None, want_std=False, packed=False, finalize=True, exact=False, **clip):
    """A stack of MORE than APGPU_MAX_STACK (512) frames on one GPU: the frames are reduced in chunks of at most `chunk`
    (default: the largest equal split not above 512), every chunk clipped against its own statistics, and the float64
    moments of the chunks accumulate in one buffer (apgpu_stack_args.moments_f64 = 2) - the single-GPU form of the
    N-shard combine (parallel.stack_nshard).  Exact for an unclipped mean; for a clipped stack the semantics are
    hierarchical (SURVEY 8(e) option ii), NOT the full-N clip.  clip: sigma, maxiters, cenfunc, stdfunc, calib, pixmask.
    packed: accumulate in the packed float64 layout (sum, count, sumsq planes) the N-shard exchange all-reduces;
    finalize=False returns the moment dict itself (a rank's share of a hierarchical stack, parallel.stack_nshard).
    Returns dict(mean, count[, std])."""
    _need_cuda(frames)
    lib = _lib.load()
    N = frames.shape[0]
    if chunk is None:
        parts = -(-N // _lib.MAX_STACK)
        chunk = -(-N // parts)
    chunk = int(min(chunk, _lib.MAX_STACK))
    shp = tuple(frames.shape[1:])
    mom = alloc_moments_f64(shp, frames.device, packed=packed)
    calib = clip.pop('calib', None)
    pixmask = clip.pop('pixmask', None)
    for k, lo in enumerate(range(0, N, chunk)):
        hi = min(N, lo + chunk)
        c = None
        if calib is not None:
            c = dict(calib)
            for key in ('exp_ratio', 'pedestal'):
                v = c.get(key)
                if v is not None and not np.isscalar(v):
                    c[key] = v[lo:hi]
        keep = []
        a, n, P, _, dev = _stack_args(frames[lo:hi], c, pixmask, keep)
        a.center = _lib.CENTER[clip.get('cenfunc', 'median')]
        a.dev = _lib.DEV[clip.get('stdfunc', 'std')]
        mi = clip.get('maxiters', 5)
        a.maxiters = -1 if mi is None else int(mi)
        sg = clip.get('sigma', 3.0)
        sl, su = clip.get('sigma_lower'), clip.get('sigma_upper')
        a.sigma_lower = float(sg if sl is None else sl)      # an explicit 0.0 is a value, not "unset"
        a.sigma_upper = float(sg if su is None else su)
        a.moments = mom['buffer'].data_ptr()
        a.moments_f64 = (3 if k == 0 else 4) if packed else (1 if k == 0 else 2)
        a.flags = (0 if want_std else _lib.STACK_MOMENTS_MEAN) | (_lib.STACK_EXACT_MOMENTS if exact else 0)   # exact: float64 clip only
        keep.append(_attach_workspace(a, P, dev))
        check(lib.apgpu_stack_sigclip(C.byref(a), _stream()))
    if not finalize:
        return mom
    out = moments_finalize(mom, want_std=want_std)
    res = dict(count=mom['count'].to(torch.int32) if packed else mom['count'])
    if want_std:
        res['mean'], res['std'] = out
    else:
        res['mean'] = out
    return res


def combine_f64(frames, sigma_lower=5.0, sigma_upper=5.0, maxiters=1, cenfunc='median', stdfunc='mad_std', form='astropy'):
    """The ccdproc.combine configuration (scripts/ap_combine_darks.py:394-420) on a FLOAT64 slab [N, ...]: one strict
    pass about the median with mad_std, float64 mean / std / count -> dict(mean_f64, std_f64, count).  float64 frames are
    never narrowed to float32 (apgpu_combine_ccdproc_f64: a correctness path, bit-identical to the oracle).
    form: 'astropy' (ccdproc >= 2.2: astropy.stats.sigma_clip) or 'legacy' (ccdproc <= 2.1), see include/apgpu.h."""
    _need_cuda(frames)
    if frames.dtype != torch.float64:
        raise TypeError('combine_f64 takes a float64 slab, got %s' % frames.dtype)
    if maxiters != 1 or cenfunc != 'median' or stdfunc != 'mad_std':
        raise ValueError('combine_f64 implements the ccdproc.combine configuration only: maxiters=1, median, mad_std')
    lib = _lib.load()
    N = frames.shape[0]
    shp = tuple(frames.shape[1:])
    P = frames[0].numel()
    if not (frames[0].is_contiguous() and (N == 1 or frames.stride(0) >= P)):
        frames = frames.contiguous()
    dev = frames.device
    res = dict(mean_f64=torch.empty(shp, dtype=torch.float64, device=dev), std_f64=torch.empty(shp, dtype=torch.float64, device=dev),
               count=torch.empty(shp, dtype=torch.int32, device=dev))
    nb = lib.apgpu_combine_ccdproc_f64_ws_bytes(N, P)
    ws = torch.empty(nb // 8, dtype=torch.float64, device=dev)
    check(lib.apgpu_combine_ccdproc_f64(_ptr(frames), N, P, frames.stride(0) if N > 1 else P, float(sigma_lower), float(sigma_upper),
                                        _lib.CCDPROC_FORM[form], _ptr(res['mean_f64']), _ptr(res['count']), _ptr(res['std_f64']), _ptr(ws), nb, _stream()))
    return res


def stack_kernel_name(n_frames, dtype='f32', calibrated=True, outputs=('mean',), median_only=False, stdfunc='std',
                      moments_mean_only=False, exact=False, maxiters=5):
    """Name of the kernel variant the library dispatches for such a stack call (apgpu_stack_kernel_name): what the
    bench line and the profiles call the dominant kernel.  Needs no device."""
    lib = _lib.load()
    a = StackArgs()
    a.frames = 0x1000                                     # placeholders: aligned, never dereferenced by the query
    a.dtype = APGPU_F32 if dtype in ('f32', torch.float32) else APGPU_U16
    a.n_frames = int(n_frames)
    a.n_pixels = 1 << 20
    a.frame_stride = 1 << 20
    if calibrated:
        a.bias = a.dark = a.nflat = a.exp_ratio = 0x1000
    a.center, a.dev, a.maxiters = 0, _lib.DEV[stdfunc], int(maxiters)
    a.sigma_lower = a.sigma_upper = 3.0
    if median_only:
        a.median = 0x1000
    for k in outputs:
        if k in ('moments_f64', 'moments_f64p'):
            a.moments, a.moments_f64 = 0x1000, (3 if k == 'moments_f64p' else 1)
        elif k == 'moments':
            a.moments = 0x1000
        else:
            setattr(a, k, 0x1000)
    a.flags = (_lib.STACK_EXACT_MOMENTS if exact else 0) | (_lib.STACK_MOMENTS_MEAN if moments_mean_only else 0)
    buf = C.create_string_buffer(256)
    check(lib.apgpu_stack_kernel_name(C.byref(a), int(bool(median_only)), buf, 256))
    return buf.value.decode()


def stack_median(frames, calib=None, pixmask=None, want_count=False):
    """np.nanmedian(cube, axis=0) (config 4), optionally fused with calibration."""
    _need_cuda(frames)
    lib = _lib.load()
    keep = []
    a, N, P, shp, dev = _stack_args(frames, calib, pixmask, keep)
    med = torch.empty(shp, dtype=torch.float32, device=dev)
    a.median = med.data_ptr()
    cnt = None
    if want_count:
        cnt = torch.empty(shp, dtype=torch.int32, device=dev)
        a.count = cnt.data_ptr()
    check(lib.apgpu_stack_median(C.byref(a), _stream()))
    return (med, cnt) if want_count else med


def alloc_moments_f64(shape, device, packed=False):
    """The float64 moment layouts of include/apgpu.h.  Default: one buffer = double sum[P], double sumsq[P], int32 count[P]
    -> dict(sum, sumsq, count, buffer) of views with the image shape.  packed: float64 [3][P] = sum, count, sumsq ->
    dict(sum, count, sumsq, buffer, prefix) with prefix = the contiguous (sum, count) planes a mean-only exchange sends."""
    shape = tuple(shape)
    P = int(np.prod(shape)) if shape else 1
    if packed:
        buf = torch.empty((3,) + shape, dtype=torch.float64, device=device)
        return dict(sum=buf[0], count=buf[1], sumsq=buf[2], buffer=buf, prefix=buf[:2], packed=True)
    buf = torch.empty(2 * P + (P + 1) // 2, dtype=torch.float64, device=device)
    return dict(sum=buf[:P].view(shape), sumsq=buf[P:2 * P].view(shape),
                count=buf[2 * P:].view(torch.int32)[:P].view(shape), buffer=buf)


def moments_finalize(moments, want_std=None, out_mean=None, want_f64=False):
    """Mean (and std) of the survivors from (all-reduced) partial moments.

    moments: the float32 tensor [2 or 3, ...] = (sum, cnt[, sumsq]) -> mean = sum / cnt in float32; no std from this
    layout (sumsq / cnt - mean^2 of float32 sums about zero cancels: ask for 'moments_f64' instead); or the dict
    stack_sigclip(..., outputs=('moments_f64',)) returns -> mean = sum / cnt, std = sqrt(sumsq / cnt - mean^2)
    evaluated in float64 and rounded once (want_f64: also return the float64 planes)."""
    lib = _lib.load()
    if want_std is None:
        want_std = isinstance(moments, dict) and moments.get('sumsq') is not None
    if isinstance(moments, dict):
        sm, sq, cnt = moments['sum'], moments.get('sumsq'), moments['count']
        _need_cuda(sm, sq, cnt)
        shp = tuple(sm.shape)
        P = sm.numel()
        packed = cnt.dtype == torch.float64                  # the packed layout carries the count as a float64 plane
        for t, dt, nm in ((sm, torch.float64, 'sum'), (sq, torch.float64, 'sumsq'), (cnt, torch.float64 if packed else torch.int32, 'count')):
            if t is not None and (t.dtype != dt or t.numel() != P or not t.is_contiguous()):
                raise TypeError('moments_f64[%r] must be a contiguous %s plane' % (nm, dt))
        if want_std and sq is None:
            raise ValueError('std needs the sumsq plane')
        if out_mean is None:
            mean = torch.empty(shp, dtype=torch.float32, device=sm.device)
        else:
            mean = out_mean
            if mean.dtype != torch.float32 or mean.numel() != P or not mean.is_contiguous():
                raise TypeError('out_mean must be a contiguous float32 tensor with one entry per pixel')
        std = torch.empty(shp, dtype=torch.float32, device=sm.device) if want_std else None
        m64 = torch.empty(shp, dtype=torch.float64, device=sm.device) if want_f64 else None
        s64 = torch.empty(shp, dtype=torch.float64, device=sm.device) if (want_f64 and want_std) else None
        if packed:
            check(lib.apgpu_moments_finalize_f64p(_ptr(sm), _ptr(cnt), _ptr(sq) if want_std else None, _ptr(mean), _ptr(std),
                                                  _ptr(m64), _ptr(s64), P, _stream()))
        else:
            check(lib.apgpu_moments_finalize_f64(_ptr(sm), _ptr(sq) if want_std else None, _ptr(cnt), _ptr(mean), _ptr(std),
                                                 _ptr(m64), _ptr(s64), P, _stream()))
        out = (mean, std) if want_std else mean
        if want_f64:
            return out, ((m64, s64) if want_std else m64)
        return out
    _need_cuda(moments)
    if want_std:
        raise ValueError("no standard deviation from float32 moments (sumsq/cnt - mean^2 cancels for CCD-range data): "
                         "use outputs=('moments_f64',)")
    moments = _f32c(moments, 'moments')
    shp = tuple(moments.shape[1:])
    P = moments[0].numel()
    if out_mean is None:
        mean = torch.empty(shp, dtype=torch.float32, device=moments.device)
    else:
        mean = out_mean
        if mean.dtype != torch.float32 or mean.numel() != P or not mean.is_contiguous():
            raise TypeError('out_mean must be a contiguous float32 tensor with one entry per pixel')
    std = torch.empty(shp, dtype=torch.float32, device=moments.device) if want_std else None
    check(lib.apgpu_moments_finalize(_ptr(moments), _ptr(mean), _ptr(std), P, _stream()))
    return (mean, std) if want_std else mean


def sigclip_global(data, sigma=3.0, sigma_lower=None, sigma_upper=None, maxiters=5):
    """A3 sigma_clipped_stats(data, sigma) with axis=None (ApFindBadPixels.py:191).

    float32 data -> numpy's float32 statistics; float64 and integer data (widened exactly) -> float64
    statistics, as numpy computes them.  Returns a float64 device tensor [10] =
    mean, median, std, lo, hi, iterations, survivors, min, max, 0."""
    _need_cuda(data)
    lib = _lib.load()
    if data.dtype == torch.float32:
        f64 = False
        data = data.contiguous()
    else:
        f64 = True
        if data.dtype == torch.uint16:
            data = (data.view(torch.int16).to(torch.int32) & 0xFFFF).to(torch.float64)
        else:
            data = data.to(torch.float64)
        data = data.contiguous()
    n = data.numel()
    ws_bytes = (lib.apgpu_sigclip_global_f64_ws_bytes if f64 else lib.apgpu_sigclip_global_ws_bytes)(n)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=data.device)
    stats = torch.empty(10, dtype=torch.float64, device=data.device)
    sl = float(sigma if sigma_lower is None else sigma_lower)
    su = float(sigma if sigma_upper is None else sigma_upper)
    fn = lib.apgpu_sigclip_global_f64 if f64 else lib.apgpu_sigclip_global_f32
    check(fn(_ptr(data), n, sl, su, -1 if maxiters is None else int(maxiters), _ptr(stats), _ptr(ws), ws_bytes, _stream()))
    return stats


def image_difference(a, b, bad1=None, bad2=None):
    """F2 ApImageDifference (ap_calc_read_noise.py:122): float64(a) - float64(b), NaN where bad1|bad2."""
    _need_cuda(a, b, bad1, bad2)
    a, b = a.contiguous(), b.contiguous()
    dt = _raw_dtype(a)
    if _raw_dtype(b) != dt or a.shape != b.shape:
        raise RuntimeError('Error, data array shapes do not match: First file=%s, second file=%s' % (tuple(a.shape), tuple(b.shape)))
    for m in (bad1, bad2):
        if m is not None and (m.dtype != torch.uint8 or m.shape != a.shape):
            raise TypeError('bad-pixel masks must be uint8 with the image shape')
    bad1 = None if bad1 is None else bad1.contiguous()
    bad2 = None if bad2 is None else bad2.contiguous()
    out = torch.empty(a.shape, dtype=torch.float64, device=a.device)
    check(_lib.load().apgpu_image_difference_f64(_ptr(a), _ptr(b), dt, _ptr(bad1), _ptr(bad2), _ptr(out), a.numel(), _stream()))
    return out


def threshold_mask(data, lothresh=0.0, hithresh=0.0, thresholds=None):
    """A4 ApFindBadPixels._generate_sigmaclip_mask (ApFindBadPixels.py:199-216).

    thresholds: optional float64 device tensor [2] read on the device instead of the two floats.
    Returns (mask uint8, nbad int64[1] device tensor)."""
    _need_cuda(data, thresholds)
    lib = _lib.load()
    data = _aligned16(_f32c(data, 'data'))
    mask = torch.empty(data.shape, dtype=torch.uint8, device=data.device)
    nbad = torch.empty(1, dtype=torch.int64, device=data.device)
    if thresholds is not None and (thresholds.dtype != torch.float64 or thresholds.numel() < 2):
        raise TypeError('thresholds must be a float64 tensor with 2 entries')
    check(lib.apgpu_threshold_mask_f32(_ptr(data), data.numel(), float(lothresh), float(hithresh), _ptr(thresholds),
                                       _ptr(mask), _ptr(nbad), _stream()))
    return mask, nbad


def mask_add_rects(mask, rects, value=2):
    """A4 overlays (ApFindBadPixels.py:90,128,154): mask[r0:r1, c0:c1] += value in place."""
    _need_cuda(mask)
    if mask.dtype != torch.uint8 or mask.dim() != 2 or not mask.is_contiguous():
        raise TypeError('mask must be a contiguous 2-D uint8 tensor')
    r = np.ascontiguousarray(np.asarray(rects, dtype=np.int32).reshape(-1, 4))
    if r.shape[0] == 0:
        return mask
    H, W = mask.shape
    if (r[:, 0] < 0).any() or (r[:, 1] > H).any() or (r[:, 2] < 0).any() or (r[:, 3] > W).any():
        raise ValueError('rectangle outside the image')
    rd = torch.from_numpy(r).to(mask.device)
    check(_lib.load().apgpu_mask_add_rects_u8(_ptr(mask), H, W, _ptr(rd), r.shape[0], int(value), _stream()))
    return mask


def fix_badpix(data, mask, deltapix=1, min_valid=4):
    """A5 ApFixBadPixels.fix_bad_pixels (ApFixBadPixels.py:292-445).

    float32 or float64 images (medians in the image's own type, as np.median); any deltapix >= 0.
    Returns (out, stats int64[3] device tensor = nbad, nfixed, nremaining)."""
    _need_cuda(data, mask)
    f64 = data.dtype == torch.float64
    data = data.contiguous() if f64 else _f32c(data, 'data')
    if data.dim() != 2:
        raise ValueError('data must be 2-D')
    if tuple(mask.shape) != tuple(data.shape):
        raise RuntimeError('Error, the shape of the input data array (%s) does not match that of the bad pixel '
                           'mask array (%s).' % (tuple(data.shape), tuple(mask.shape)))
    m8 = (mask != 0).to(torch.uint8).contiguous() if mask.dtype != torch.uint8 else mask.contiguous()
    out = torch.empty_like(data)
    stats = torch.empty(3, dtype=torch.int64, device=data.device)
    fn = _lib.load().apgpu_fix_badpix_f64 if f64 else _lib.load().apgpu_fix_badpix_f32
    check(fn(_ptr(data), _ptr(m8), data.shape[0], data.shape[1], int(deltapix), int(min_valid), _ptr(out), _ptr(stats), _stream()))
    return out, stats


def imarith(a, op, b):
    """A8 ApImArith op block (ApImArith.py:320-333): a (op) b, b a tensor or a python float."""
    _need_cuda(a)
    a = a.contiguous()
    opi = _lib.OPS[op]
    b_is_t = torch.is_tensor(b)
    if a.dtype == torch.float64 or (b_is_t and b.dtype == torch.float64 and a.dtype == torch.float32):
        # float64 somewhere: computed in float64, stored in a's dtype (numpy, out=zeros_like(data1))
        out = torch.empty_like(a)
        adt = _dtype_tag(a, 'a')
        if b_is_t:
            _need_cuda(b)
            if b.shape != a.shape:
                raise RuntimeError('Error, the dimension of the second data array does not match the first.')
            b = b.contiguous()
            check(_lib.load().apgpu_imarith_f64(_ptr(a), adt, _ptr(b), _dtype_tag(b, 'b'), 0.0, opi, _ptr(out), a.numel(), _stream()))
        else:
            check(_lib.load().apgpu_imarith_f64(_ptr(a), adt, None, APGPU_F64, float(b), opi, _ptr(out), a.numel(), _stream()))
        return out
    dt = _raw_dtype(a)
    a = _aligned16(a)
    out = torch.empty_like(a)
    if torch.is_tensor(b):
        _need_cuda(b)
        if b.shape != a.shape:
            raise RuntimeError('Error, the dimension of the second data array does not match the first.')
        if _raw_dtype(b) != dt:
            raise TypeError('operands must have the same dtype')
        b = _aligned16(b)
        check(_lib.load().apgpu_imarith(_ptr(a), _ptr(b), 0.0, opi, dt, _ptr(out), a.numel(), _stream()))
    else:
        check(_lib.load().apgpu_imarith(_ptr(a), None, float(b), opi, dt, _ptr(out), a.numel(), _stream()))
    return out


def bayer_split(raw, pattern=(0, 1, 3, 2), black=None):
    """A9 RawConv split geometry (RawConv.py:111-128): four full-size planes [4,H,W] (R, G1, B, G2)."""
    _need_cuda(raw)
    if _raw_dtype(raw) != APGPU_U16 or raw.dim() != 2:
        raise TypeError('raw must be a 2-D uint16 tensor')
    raw = raw.contiguous()
    H, W = raw.shape
    planes = torch.empty((4, H, W), dtype=raw.dtype, device=raw.device)
    pat = (C.c_int32 * 4)(*[int(x) for x in pattern])
    blk = (C.c_int32 * 4)(*[int(x) for x in black]) if black is not None else None
    check(_lib.load().apgpu_bayer_split_u16(_ptr(raw), H, W, pat, blk, _ptr(planes), _stream()))
    return planes


def bayer_flat_normalize(flat):
    """Per-channel flat normalisation for a Bayer mosaic (config 4): every 2x2 cell position (R, G1, G2, B)
    is normalised by the nanmean of its own quarter-plane with the A1 kernel, i.e. four applications of
    ApCalibrate._generate_flat (ApCalibrate.py:166-190) on the planes RawConv.split separates
    (RawConv.py:111-128).  Returns (nflat[H,W], norms[2,2] device tensor)."""
    _need_cuda(flat)
    flat = _f32c(flat, 'flat')
    if flat.dim() != 2 or flat.shape[0] % 2 or flat.shape[1] % 2:
        raise ValueError('a Bayer mosaic needs an even number of rows and columns')
    nflat = torch.empty_like(flat)
    norms = torch.empty((2, 2), dtype=torch.float32, device=flat.device)
    for r0 in (0, 1):
        for c0 in (0, 1):
            plane = flat[r0::2, c0::2].contiguous()
            npl, norm = flat_normalize(plane)
            nflat[r0::2, c0::2] = npl
            norms[r0, c0] = norm[0]
    return nflat, norms


# ---------------------------------------------------------------------------------------------------
# F3: registration resample (what the reference runs SWarp for, scripts/resample_all.sh:123-131, 330-342)
# ---------------------------------------------------------------------------------------------------
_LUT_CACHE = {}


def lanczos3_table(n_phases=1024, device='cuda'):
    """[n_phases + 1, 6] float32 device tensor: row p = normalised Lanczos-3 weights of the taps
    floor(x)-2 .. floor(x)+3 for a fractional offset t = p / n_phases (tap k at distance d = k - 2 - t,
    L(d) = sinc(d) sinc(d/3) for |d| < 3).  Built once per (n_phases, device) in float64 on the host."""
    import numpy as np
    key = (int(n_phases), str(device))
    if key not in _LUT_CACHE:
        t = np.arange(n_phases + 1, dtype=np.float64)[:, None] / n_phases
        d = np.arange(6, dtype=np.float64)[None, :] - 2.0 - t
        w = np.sinc(d) * np.sinc(d / 3.0)
        w[np.abs(d) >= 3.0] = 0.0
        w[np.abs(w) < 1e-12] = 0.0              # np.sinc(integer) is ~1e-17, not 0: whole-pixel offsets are exact copies
        w /= w.sum(axis=1, keepdims=True)
        _LUT_CACHE[key] = torch.from_numpy(np.ascontiguousarray(w.astype(np.float32))).to(device)
    return _LUT_CACHE[key]


def _resample_args(frames, affines, fscale, mask, out_shape, n_phases):
    """Shared argument preparation of the resample entry points: frames [N,H,W] float32 contiguous, transforms as a float64 device
    tensor ([N,6] or per-tile [N,ty,tx,6]), flux scales, mask as uint8, the weight table -> (frames, N, H, W, h, w, aff, per_tile, fs, mk, lut)."""
    _need_cuda(frames)
    frames = _f32c(frames, 'frames')
    if frames.dim() == 2:
        frames = frames[None]
    if frames.dim() != 3:
        raise ValueError('frames must be [N,H,W] or [H,W]')
    N, H, W = frames.shape
    dev = frames.device
    h, w = (H, W) if out_shape is None else (int(out_shape[0]), int(out_shape[1]))
    aff = torch.as_tensor(affines, dtype=torch.float64)
    per_tile = aff.dim() == 4
    if per_tile:
        ty, tx = (h + 15) // 16, (w + 63) // 64
        if tuple(aff.shape) != (N, ty, tx, 6):
            raise ValueError('per-tile affines must be [N, %d, %d, 6] for a %d x %d output' % (ty, tx, h, w))
    else:
        aff = aff.reshape(-1, 6)
        if aff.shape[0] == 1 and N > 1:
            aff = aff.expand(N, 6)
        if aff.shape[0] != N:
            raise ValueError('affines must hold one 2x3 transform per frame')
    aff = aff.contiguous().to(dev)
    fs = None
    if fscale is not None:
        fs = torch.as_tensor(fscale, dtype=torch.float32).reshape(-1)
        if fs.numel() == 1 and N > 1:
            fs = fs.expand(N)
        if fs.numel() != N:
            raise ValueError('fscale must hold one value per frame')
        fs = fs.contiguous().to(dev)
    mk = None
    if mask is not None:
        _need_cuda(mask)
        if tuple(mask.shape) != (H, W):
            raise ValueError('mask must be [H,W]')
        mk = mask.contiguous() if mask.dtype == torch.uint8 else (mask != 0).to(torch.uint8)
    return frames, N, H, W, h, w, aff, per_tile, fs, mk, lanczos3_table(n_phases, dev)


def resample_affine(frames, affines, fscale=None, mask=None, out_shape=None, n_phases=1024, out=None, weight=True,
                    conserve_flux=False):
    """Affine Lanczos-3 resample of [N,H,W] (or [H,W]) float32 frames onto a common grid.

    affines: [N,6] float64 (tensor / array / nested list): xin = A0*x + A1*y + A2, yin = A3*x + A4*y + A5 maps an
    OUTPUT pixel (x = column, y = row) to INPUT coordinates; or [N, tiles_y, tiles_x, 6] with one transform per
    16 x 64 output tile (wcs.tile_affines: a piecewise-affine TAN -> TAN registration).  fscale: per-frame flux scale (SWarp's FSCALE,
    1/EXPTIME in resample_all.sh:298) or None.  conserve_flux: also scale by the local pixel-area ratio |det A| (SWarp's
    FSCALASTRO_TYPE VARIABLE, resample_all.sh:129).  mask: [H,W] uint8, non-zero = bad pixel, shared by all frames.
    Returns (resampled [N,h,w] float32 with NaN where undefined, weight uint8 [N,h,w] or None).  The co-add is
    stack_median / stack_sigclip on the result (both skip NaN)."""
    frames, N, H, W, h, w, aff, per_tile, fs, mk, lut = _resample_args(frames, affines, fscale, mask, out_shape, n_phases)
    dev = frames.device
    if out is None:
        out = torch.empty((N, h, w), dtype=torch.float32, device=dev)
    elif tuple(out.shape) != (N, h, w) or out.dtype != torch.float32 or not out.is_contiguous():
        raise ValueError('out must be a contiguous float32 [N,h,w] tensor')
    wt = torch.empty((N, h, w), dtype=torch.uint8, device=dev) if weight else None
    check(_lib.load().apgpu_resample_affine_f32(_ptr(frames), N, H, W, _ptr(mk) if mk is not None else None, _ptr(aff),
                                                int(per_tile), int(bool(conserve_flux)), _ptr(fs) if fs is not None else None, _ptr(lut), int(n_phases), _ptr(out),
                                                _ptr(wt) if wt is not None else None, h, w, _stream()))
    return out, wt


_fused_ws = {}


def resample_stack_sigclip(frames, affines, fscale=None, mask=None, out_shape=None, n_phases=1024, conserve_flux=False,
                           sigma=3.0, sigma_lower=None, sigma_upper=None, maxiters=5, cenfunc='median', outputs=('mean',),
                           exact=False, moments_mean_only=False):
    """resample_affine + stack_sigclip in ONE launch (apgpu_resample_stack_sigclip): the co-add of registered frames without
    the resampled slab - every output pixel's N Lanczos-3 values go straight into the clip (scripts/resample_all.sh:330-342,
    one SWarp call there).  frames [N <= 16, H, W] float32; affines / fscale / mask / out_shape / n_phases / conserve_flux as
    resample_affine; the clip's arguments as stack_sigclip (stdfunc 'std'); outputs among 'mean', 'count', 'moments',
    'moments_f64', 'moments_f64p'.  Same survivors as the two-step form; the mean within the float32 fast path's rounding."""
    frames, N, H, W, h, w, aff, per_tile, fs, mk, lut = _resample_args(frames, affines, fscale, mask, out_shape, n_phases)
    if N > 16:
        raise ValueError('resample_stack_sigclip takes up to 16 frames per call (resample_affine + stack_sigclip beyond)')
    dev = frames.device
    lib = _lib.load()
    a = StackArgs()
    a.frames = frames.data_ptr()
    a.dtype = _raw_dtype(frames)
    a.n_frames = N
    a.n_pixels = h * w
    a.frame_stride = H * W
    a.center = _lib.CENTER[cenfunc]
    a.dev = _lib.DEV['std']
    a.maxiters = -1 if maxiters is None else int(maxiters)
    a.sigma_lower = float(sigma if sigma_lower is None else sigma_lower)
    a.sigma_upper = float(sigma if sigma_upper is None else sigma_upper)
    shp = (h, w)
    res = {}
    for k in outputs:
        if k == 'mean':
            res[k] = torch.empty(shp, dtype=torch.float32, device=dev)
        elif k == 'count':
            res[k] = torch.empty(shp, dtype=torch.int32, device=dev)
        elif k == 'moments':
            if a.moments:
                raise ValueError('ask for one moment layout')
            res[k] = torch.empty((3,) + shp, dtype=torch.float32, device=dev)
            a.moments = res[k].data_ptr()
            continue
        elif k in ('moments_f64', 'moments_f64p'):
            if a.moments:
                raise ValueError('ask for one moment layout')
            res[k] = alloc_moments_f64(shp, dev, packed=(k == 'moments_f64p'))
            a.moments = res[k]['buffer'].data_ptr()
            a.moments_f64 = 3 if k == 'moments_f64p' else 1
            continue
        else:
            raise ValueError('unknown output %r (mean, count, moments, moments_f64, moments_f64p)' % (k,))
        setattr(a, k, res[k].data_ptr())
    a.flags = (_lib.STACK_EXACT_MOMENTS if exact else 0) | (_lib.STACK_MOMENTS_MEAN if moments_mean_only else 0)
    nb = lib.apgpu_resample_stack_ws_bytes(N, H, W, h, w, int(mk is not None))
    key = (dev.index, int(torch.cuda.current_stream(dev).cuda_stream))
    ws = _fused_ws.get(key)
    if ws is None or ws.numel() < nb:
        ws = torch.empty(nb + 64, dtype=torch.uint8, device=dev)        # (torch allocations are 512-byte aligned)
        _fused_ws[key] = ws
    check(lib.apgpu_resample_stack_sigclip(C.byref(a), H, W, _ptr(mk) if mk is not None else None, _ptr(aff), int(per_tile),
                                           int(bool(conserve_flux)), _ptr(fs) if fs is not None else None, _ptr(lut), int(n_phases),
                                           h, w, _ptr(ws), ws.numel(), _stream()))
    return res


def oversampled_affines(affines, oversampling, out_shape):
    """The transforms of the n-times finer grid whose n x n blocks are the output pixels: fine pixel (u, v) has its centre at
    output coordinates ((u + 0.5) / n - 0.5, (v + 0.5) / n - 0.5).  Single transform per frame only ([N, 6])."""
    n = int(oversampling)
    aff = torch.as_tensor(affines, dtype=torch.float64).reshape(-1, 6).clone()
    off = 0.5 / n - 0.5
    fine = aff.clone()
    fine[:, 0] = aff[:, 0] / n
    fine[:, 1] = aff[:, 1] / n
    fine[:, 2] = aff[:, 2] + (aff[:, 0] + aff[:, 1]) * off
    fine[:, 3] = aff[:, 3] / n
    fine[:, 4] = aff[:, 4] / n
    fine[:, 5] = aff[:, 5] + (aff[:, 3] + aff[:, 4]) * off
    return fine, (int(out_shape[0]) * n, int(out_shape[1]) * n)


def block_mean(fine, oversampling):
    """[n*h, n*w] float32 -> [h, w]: mean of the n x n sub-samples of every output pixel (NaN if any is NaN)."""
    _need_cuda(fine)
    fine = _f32c(fine, 'fine')
    n = int(oversampling)
    hf, wf = fine.shape
    if hf % n or wf % n:
        raise ValueError('the fine grid must be a whole multiple of the oversampling factor')
    out = torch.empty((hf // n, wf // n), dtype=torch.float32, device=fine.device)
    check(_lib.load().apgpu_block_mean_f32(_ptr(fine), hf // n, wf // n, n, _ptr(out), _stream()))
    return out


def _oversampling_args(frames, affines, oversampling, fscale, out_shape, conserve_flux, fine_affines, tile_scale):
    frames = _f32c(frames, 'frames')
    if frames.dim() == 2:
        frames = frames[None]
    N, H, W = frames.shape
    n = int(oversampling)
    if n < 1 or n > 16:
        raise ValueError('oversampling must be 1..16')
    h, w = (H, W) if out_shape is None else (int(out_shape[0]), int(out_shape[1]))
    if fine_affines is None:
        fine_aff, _ = oversampled_affines(affines, n, (h, w))
        if fine_aff.shape[0] == 1 and N > 1:
            fine_aff = fine_aff.expand(N, 6)
        if fine_aff.shape[0] != N:
            raise ValueError('affines must hold one 2x3 transform per frame')
    else:
        fine_aff = torch.as_tensor(fine_affines, dtype=torch.float64)
        ty, tx = -(-h * n // (16 * tile_scale)), -(-w * n // (64 * tile_scale))
        if tuple(fine_aff.shape) != (N, ty, tx, 6):
            raise ValueError('per-tile fine affines must be [N, %d, %d, 6] for a %d x %d output at oversampling %d' % (ty, tx, h, w, n))
    fs = torch.ones(N, dtype=torch.float32) if fscale is None else torch.as_tensor(fscale, dtype=torch.float32).reshape(-1).cpu()
    if fs.numel() == 1 and N > 1:
        fs = fs.expand(N)
    if fs.numel() != N:
        raise ValueError('fscale must hold one value per frame')
    # |det| of the fine transform is 1 / n^2 of the output pixel's: the block MEAN of flux-conserving fine values times n^2
    # is the output pixel's value, so the scale goes into the per-frame flux factor
    area = float(n * n) if conserve_flux else 1.0
    return frames, N, H, W, n, h, w, fine_aff, (fs.to(torch.float64) * area).to(torch.float32)


def resample_oversampled(frames, affines, oversampling, fscale=None, mask=None, out_shape=None, n_phases=1024, conserve_flux=False,
                         fine_affines=None):
    """SWarp's OVERSAMPLING n (resample_all.sh:112, 339): every output pixel is the mean of n x n Lanczos-3 interpolations at
    the centres of its sub-pixels, in ONE kernel (apgpu_resample_oversampled_f32: the sub-samples of a pixel are evaluated and
    averaged in registers; the n-times finer image never exists).  `fine_affines`: transforms of the FINE grid per OUTPUT
    tile ([N, ceil(h/16), ceil(w/64), 6], e.g. wcs.tile_affines(fine_wcs, in_wcs, fine_shape, tile_scale=n)) instead of one
    transform per frame.  Returns [N, h, w] float32, NaN where undefined."""
    _need_cuda(frames)
    frames, N, H, W, n, h, w, fine_aff, fs = _oversampling_args(frames, affines, oversampling, fscale, out_shape, conserve_flux,
                                                                 fine_affines, int(oversampling))
    dev = frames.device
    mk = None
    if mask is not None:
        _need_cuda(mask)
        if tuple(mask.shape) != (H, W):
            raise ValueError('mask must be [H,W]')
        mk = mask.contiguous() if mask.dtype == torch.uint8 else (mask != 0).to(torch.uint8)
    fine_aff = fine_aff.contiguous().to(dev)
    fs = fs.contiguous().to(dev)
    lut = lanczos3_table(n_phases, dev)
    out = torch.empty((N, h, w), dtype=torch.float32, device=dev)
    check(_lib.load().apgpu_resample_oversampled_f32(_ptr(frames), N, H, W, _ptr(mk) if mk is not None else None, _ptr(fine_aff),
                                                     int(fine_aff.dim() == 4), int(bool(conserve_flux)), _ptr(fs), _ptr(lut), int(n_phases), n,
                                                     _ptr(out), None, h, w, _stream()))
    return out


def resample_oversampled_two_step(frames, affines, oversampling, fscale=None, mask=None, out_shape=None, n_phases=1024,
                                  conserve_flux=False, fine_affines=None):
    """The two-step form of resample_oversampled (rounds 2-3a): one frame at a time through the n-times finer grid (n^2 x the
    output size in HBM: apgpu_resample_affine_f32), then apgpu_block_mean_f32.  Bit-identical to the fused kernel for one
    transform per frame; kept as its cross-check and for the benchmark.  `fine_affines` here is per tile of the FINE grid."""
    frames, N, H, W, n, h, w, fine_aff, fs = _oversampling_args(frames, affines, oversampling, fscale, out_shape, conserve_flux,
                                                                 fine_affines, 1)
    out = torch.empty((N, h, w), dtype=torch.float32, device=frames.device)
    for i in range(N):
        fine, _ = resample_affine(frames[i:i + 1], fine_aff[i:i + 1], fscale=fs[i:i + 1], mask=mask, out_shape=(h * n, w * n),
                                  n_phases=n_phases, weight=False, conserve_flux=conserve_flux)
        check(_lib.load().apgpu_block_mean_f32(_ptr(fine), h, w, n, _ptr(out[i]), _stream()))
    return out


def weighted_mean(slab, weights):
    """COMBINE_TYPE WEIGHTED with one weight per frame: (sum w_i x_i / sum w_i over the finite x_i, sum of those w_i) per pixel:
    float32 [H, W] each; NaN / 0 where no frame contributes."""
    _need_cuda(slab)
    slab = _f32c(slab, 'slab')
    N = slab.shape[0]
    wts = torch.as_tensor(weights, dtype=torch.float32).reshape(-1).to(slab.device).contiguous()
    if wts.numel() != N:
        raise ValueError('weights must hold one value per frame')
    if not bool(torch.isfinite(wts).all()) or not bool((wts > 0).all()):
        raise ValueError('weights must be finite and positive')
    P = slab[0].numel()
    mean = torch.empty(slab.shape[1:], dtype=torch.float32, device=slab.device)
    wsum = torch.empty_like(mean)
    check(_lib.load().apgpu_weighted_mean_f32(_ptr(slab), N, P, _ptr(wts), _ptr(mean), _ptr(wsum), _stream()))
    return mean, wsum


def background_weights(frames, fscale=None, sigma=3.0, maxiters=5):
    """SWarp's weights without weight maps (WEIGHT_TYPE NONE): a frame counts with the inverse variance of its flux-scaled
    background noise, w_i = 1 / (fscale_i * sigma_i)^2.  sigma_i here = the sigma-clipped standard deviation of the whole
    frame (the A3 kernels); SWarp measures it on its own background mesh.  Returns float64 numpy [N]."""
    import numpy as np
    N = frames.shape[0]
    fs = np.ones(N) if fscale is None else np.broadcast_to(np.asarray(fscale, dtype=np.float64).reshape(-1), (N,))
    sd = np.array([float(sigclip_global(frames[i], sigma=sigma, maxiters=maxiters)[2].item()) for i in range(N)])
    if not np.all(np.isfinite(sd)) or np.any(sd <= 0):
        raise ValueError('a frame has no measurable background noise (constant or empty): give explicit weights')
    return 1.0 / (fs * sd) ** 2


def coadd(frames, affines, fscale=None, mask=None, out_shape=None, combine='MEDIAN', sigma=3.0, maxiters=5, n_phases=1024,
          conserve_flux=False, oversampling=1, weights=None, fine_affines=None, fused=False):
    """Resample + combine: SWarp's COMBINE_TYPE MEDIAN / AVERAGE / WEIGHTED / SUM (resample_all.sh:60-73 add modes) plus
    CLIPPED (sigma-clipped mean, median-centred); oversampling = SWarp's OVERSAMPLING.  WEIGHTED takes one weight per frame
    (default: background_weights).  Returns dict(image, count[, weight]) - count = frames contributing, weight = the sum
    of their weights (WEIGHTED only).
    fused=True (CLIPPED / AVERAGE, up to 16 frames, oversampling 1): one launch, no [N, h, w] slab of resampled frames in
    memory (resample_stack_sigclip: same survivors; 6.2 against 5.1 ms for C5's share on one MI355X, but 4 N h w bytes less HBM -
    DESIGN 4.4d)."""
    combine = combine.upper()
    if fused and int(oversampling) == 1 and combine in ('CLIPPED', 'AVERAGE') and frames.dim() == 3 and frames.shape[0] <= 16:
        kw = dict(sigma=sigma, maxiters=maxiters) if combine == 'CLIPPED' else dict(sigma=1e30, maxiters=1, cenfunc='mean')
        r = resample_stack_sigclip(frames, affines, fscale=fscale, mask=mask, out_shape=out_shape, n_phases=n_phases,
                                   conserve_flux=conserve_flux, outputs=('mean', 'count'), **kw)
        return dict(image=r['mean'], count=r['count'])
    if int(oversampling) > 1:
        res = resample_oversampled(frames, affines, oversampling, fscale=fscale, mask=mask, out_shape=out_shape, n_phases=n_phases,
                                   conserve_flux=conserve_flux, fine_affines=fine_affines)
    else:
        res, _ = resample_affine(frames, affines, fscale=fscale, mask=mask, out_shape=out_shape, n_phases=n_phases, weight=False,
                                 conserve_flux=conserve_flux)
    if combine == 'MEDIAN':
        med, cnt = stack_median(res, want_count=True)
        return dict(image=med, count=cnt)
    if combine == 'WEIGHTED':
        if weights is None:
            fr = frames if frames.dim() == 3 else frames[None]
            weights = background_weights(fr, fscale)
        mean, wsum = weighted_mean(res, weights)
        cnt = stack_sigclip(res, sigma=1e30, maxiters=1, cenfunc='mean', outputs=('count',))['count']
        return dict(image=mean, count=cnt, weight=wsum)
    if combine in ('AVERAGE', 'SUM'):
        # one pass with bounds nothing can exceed = np.nanmean / np.nansum along N
        r = stack_sigclip(res, sigma=1e30, maxiters=1, cenfunc='mean',
                          outputs=('mean', 'count') if combine != 'SUM' else ('moments', 'count'))
        if combine == 'SUM':
            return dict(image=r['moments'][0], count=r['count'])
        return dict(image=r['mean'], count=r['count'])
    if combine == 'CLIPPED':
        r = stack_sigclip(res, sigma=sigma, maxiters=maxiters, outputs=('mean', 'count'))
        return dict(image=r['mean'], count=r['count'])
    raise ValueError("combine must be one of MEDIAN, AVERAGE, WEIGHTED, SUM, CLIPPED")


# ---------------------------------------------------------------------------------------------------
# F4: sky-background mesh (core/ApMeasureBackground.py:142-175, 382-415; photutils restated, parity unpinned)
# ---------------------------------------------------------------------------------------------------
def source_mask(above, min_pixels=5, dilate_size=13):
    """detect_sources(npixels=min_pixels, 8-connectivity) + make_source_mask(size=dilate_size) on a uint8 map of the pixels
    above the detection threshold.  Returns (mask uint8 [H,W], nsources int64[1] device tensor)."""
    _need_cuda(above)
    if above.dtype != torch.uint8 or above.dim() != 2:
        raise TypeError('above must be a 2-D uint8 tensor')
    above = above.contiguous()
    lib = _lib.load()
    H, W = above.shape
    ws_bytes = lib.apgpu_source_mask_ws_bytes(H, W)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=above.device)
    out = torch.empty_like(above)
    nsrc = torch.empty(1, dtype=torch.int64, device=above.device)
    check(lib.apgpu_source_mask_u8(_ptr(above), H, W, int(min_pixels), int(dilate_size), _ptr(out), _ptr(nsrc), _ptr(ws), ws_bytes,
                                   _stream()))
    return out, nsrc


def box_clipped_stats(data, mask, box_height, box_width, sigma=3.0, maxiters=5):
    """Background2D's per-box SigmaClip(sigma, maxiters) + median / std: float64 device tensor [ny, nx, 4] =
    median, std, survivors, pixels masked before clipping (boxes sticking out of the image are padded with masked pixels)."""
    _need_cuda(data, mask)
    data = _f32c(data, 'data')
    if data.dim() != 2:
        raise ValueError('data must be 2-D')
    if mask is not None:
        if mask.dtype != torch.uint8 or tuple(mask.shape) != tuple(data.shape):
            raise TypeError('mask must be uint8 with the image shape')
        mask = mask.contiguous()
    H, W = data.shape
    ny, nx = -(-H // int(box_height)), -(-W // int(box_width))
    out = torch.empty((ny, nx, 4), dtype=torch.float64, device=data.device)
    check(_lib.load().apgpu_box_clipped_stats_f32(_ptr(data), _ptr(mask), H, W, int(box_height), int(box_width), float(sigma),
                                                  int(maxiters), _ptr(out), _stream()))
    return out


def spline_zoom(coef, zoom_y, zoom_x, height, width, vmin, vmax):
    """scipy.ndimage.zoom(mesh, (zoom_y, zoom_x), order=3, mode='reflect', grid_mode=True)[:height, :width] from prefiltered
    B-spline coefficients coef [ny, nx] (float64 device tensor), clipped to [vmin, vmax]; float64 [height, width]."""
    _need_cuda(coef)
    if coef.dtype != torch.float64 or coef.dim() != 2:
        raise TypeError('coef must be a 2-D float64 tensor')
    coef = coef.contiguous()
    ny, nx = coef.shape
    out = torch.empty((int(height), int(width)), dtype=torch.float64, device=coef.device)
    check(_lib.load().apgpu_spline_zoom_f64(_ptr(coef), ny, nx, int(zoom_y), int(zoom_x), int(height), int(width), float(vmin),
                                            float(vmax), _ptr(out), _stream()))
    return out


# ---------------------------------------------------------------------------------------------------
# F4: L.A.Cosmic (core/ApFixCosmicRays.py:267-295 -> ccdproc -> astroscrappy; restated, parity unpinned)
# ---------------------------------------------------------------------------------------------------
def sepmedfilt(data, size):
    """astroscrappy's separable median filter: median of `size` (5, 7, 9) along rows, then along columns, borders copied."""
    _need_cuda(data)
    data = _f32c(data, 'data')
    out = torch.empty_like(data)
    ws = torch.empty(data.numel() * 4, dtype=torch.uint8, device=data.device)
    check(_lib.load().apgpu_sepmedfilt_f32(_ptr(data), data.shape[0], data.shape[1], int(size), _ptr(out), _ptr(ws), ws.numel(), _stream()))
    return out


def gauss_psf_kernel(fwhm, size=7, device='cuda'):
    """astroscrappy's gausskernel(psffwhm, kernsize): normalised float32 Gaussian, [size, size] device tensor."""
    x = np.tile(np.arange(size) - size // 2, (size, 1)).astype(np.float64)
    y = x.T.copy()
    sigma2 = fwhm * fwhm / 2.35482 / 2.35482
    k = np.exp(-0.5 * (x * x + y * y) / sigma2).astype(np.float32)
    return torch.from_numpy((k / k.sum()).astype(np.float32)).to(device)


def lacosmic(data_electrons, inmask=None, sigclip=4.5, sigfrac=0.3, objlim=5.0, readnoise=12.0, satlevel=65535.0, niter=6,
             psffwhm=3.5, fsmode='convolve'):
    """astroscrappy.detect_cosmics on a float32 image in electrons (device tensor): returns (cleaned float32 tensor,
    crmask uint8 tensor, iterations run).  Non-finite pixels must have been zeroed and put into inmask by the caller.
    One host read of the per-iteration cosmic-ray count decides whether to go on (astroscrappy stops at 0)."""
    _need_cuda(data_electrons, inmask)
    lib = _lib.load()
    clean = _f32c(data_electrons, 'data').clone()
    H, W = clean.shape
    if inmask is not None and (inmask.dtype != torch.uint8 or tuple(inmask.shape) != (H, W)):
        raise TypeError('inmask must be uint8 with the image shape')
    ws_bytes = lib.apgpu_lacosmic_ws_bytes(H, W)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=clean.device)
    mask = torch.empty((H, W), dtype=torch.uint8, device=clean.device)
    check(lib.apgpu_lacosmic_satmask(_ptr(clean), _ptr(inmask.contiguous()) if inmask is not None else None, H, W, float(satlevel),
                                     _ptr(mask), _ptr(ws), ws_bytes, _stream()))
    # background level for cosmic rays without a good neighbour: np.median of the unmasked pixels (A3 kernels, no clipping)
    masked = torch.where(mask != 0, torch.full_like(clean, float('nan')), clean)
    bkg = float(sigclip_global(masked, sigma=1e30, maxiters=1)[1].item())
    if bkg != bkg:
        bkg = 0.0
    psfk = gauss_psf_kernel(psffwhm, 7, clean.device) if fsmode == 'convolve' else None
    crmask = torch.zeros((H, W), dtype=torch.uint8, device=clean.device)
    ncr = torch.zeros(1, dtype=torch.int64, device=clean.device)
    it = 0
    for it in range(1, int(niter) + 1):
        check(lib.apgpu_lacosmic_iterate(_ptr(clean), _ptr(mask), _ptr(crmask), H, W, float(sigclip), float(sigfrac), float(objlim),
                                         float(readnoise), _ptr(psfk), bkg, _ptr(ncr), _ptr(ws), ws_bytes, _stream()))
        if int(ncr.item()) == 0:
            break
    return clean, crmask, it
