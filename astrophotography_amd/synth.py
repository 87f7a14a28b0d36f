"""Deterministic synthetic CCD data generated directly in HBM.

Used by bench.py and the tests; there is no network for real frames.  Pixel model:
  bias ~ N(1000, 5); dark ~ N(20, 3) with 0.02 % hot pixels U(2000, 6000) and 0.005 % cold pixels -300;
  flat ~ N(30000, 300) * (1 - 0.3 r^2) with one pixel exactly 0; sky = 500 + 50 x / W plus Gaussian stars
  (sigma 2 px, peak U(100, 20000)); raw = clip(bias + e*dark + nflat*(sky + stars) + noise, 0, 65535) with
  noise sigma sqrt(signal) + 12 and 0.1 % cosmic-ray pixels += U(500, 5000) per frame; e = 120/300.
Seeds: 1000 * config_id + frame index (masters use frame index 999).
"""
import torch

EXP_RATIO = 120.0 / 300.0


def _gen(device, seed):
    g = torch.Generator(device=device)
    g.manual_seed(int(seed))
    return g


def make_masters(H, W, config_id=2, device='cuda'):
    g = _gen(device, 1000 * config_id + 999)
    bias = torch.randn((H, W), generator=g, device=device) * 5 + 1000
    dark = torch.randn((H, W), generator=g, device=device) * 3 + 20
    u = torch.rand((H, W), generator=g, device=device)
    hot = u < 0.0002
    dark = torch.where(hot, torch.rand((H, W), generator=g, device=device) * 4000 + 2000, dark)
    dark = torch.where((u > 0.5) & (u < 0.50005), torch.full_like(dark, -300.0), dark)
    yy = (torch.arange(H, device=device, dtype=torch.float32) - H / 2).view(H, 1)
    xx = (torch.arange(W, device=device, dtype=torch.float32) - W / 2).view(1, W)
    r2 = (yy * yy + xx * xx) / (H * H / 4 + W * W / 4)
    flat = (torch.randn((H, W), generator=g, device=device) * 300 + 30000) * (1 - 0.3 * r2)
    flat[H // 3, W // 5] = 0.0
    # star field: sparse peaks blurred by a separable sigma = 2 px Gaussian
    nstars = max(1, int(200 * (H * W) / (512 * 512)))
    stars = torch.zeros((H, W), device=device)
    idx = torch.randint(0, H * W, (nstars,), generator=g, device=device)
    peak = torch.rand((nstars,), generator=g, device=device) * 19900 + 100
    stars.view(-1).index_put_((idx,), peak * (2 * 3.14159265 * 4.0), accumulate=True)
    k = torch.exp(-0.5 * (torch.arange(-6, 7, device=device, dtype=torch.float32) / 2.0) ** 2)
    k = k / k.sum()
    pad = torch.nn.functional.pad
    s = pad(stars, (6, 6, 0, 0))
    stars = sum(k[i] * s[:, i:i + W] for i in range(13))
    s = pad(stars, (0, 0, 6, 6))
    stars = sum(k[i] * s[i:i + H, :] for i in range(13))
    sky = 500 + 50 * torch.arange(W, device=device, dtype=torch.float32).view(1, W) / W + stars
    return dict(bias=bias.contiguous(), dark=dark.contiguous(), flat=flat.contiguous(), scene=sky.expand(H, W).contiguous())


def make_frames(N, masters, nflat, config_id=2, dtype=torch.float32, first_frame=0, out=None, scenes=None):
    """raw[N,H,W] (float32 or uint16 holding integer ADU) for the masters returned by make_masters.
    scenes: None (every frame images masters['scene'] at the same place) or a callable f -> [H,W] scene of frame f (a dithered
    sequence: the sky as frame f saw it, e.g. the scene warped by the inverse of the frame's registration transform)."""
    bias, dark, scene = masters['bias'], masters['dark'], masters['scene']
    H, W = bias.shape
    device = bias.device
    if out is None:
        out = torch.empty((N, H, W), dtype=dtype, device=device)
    signal = bias + EXP_RATIO * dark + nflat * scene
    sig_noise = signal.clamp_min(0).sqrt() + 12
    for f in range(N):
        g = _gen(device, 1000 * config_id + first_frame + f)
        if scenes is not None:
            signal = bias + EXP_RATIO * dark + nflat * scenes(f)
            sig_noise = signal.clamp_min(0).sqrt() + 12
        fr = signal + torch.randn((H, W), generator=g, device=device) * sig_noise
        cr = torch.rand((H, W), generator=g, device=device) < 0.001
        fr = fr + cr * (torch.rand((H, W), generator=g, device=device) * 4500 + 500)
        fr = fr.clamp_(0, 65535)
        if dtype == torch.float32:
            out[f] = fr
        else:
            out[f] = fr.round_().to(torch.int32).to(torch.uint16) if hasattr(torch, 'uint16') else fr
    return out


def make_sky_frame(H, W, seed=7, nstars=None, ncr=None, sky=400.0, read_noise=8.0, device='cuda'):
    """A calibrated light frame in electrons for the per-frame F4 steps (background mesh, cosmic rays): sky level with a
    gentle gradient, Gaussian-PSF stars (sigma 1.5 px), shot + read noise, and `ncr` cosmic-ray hits (single pixels and short
    tracks, sharper than the PSF).  Returns (frame float32 [H, W], cosmic-ray truth mask bool [H, W])."""
    g = _gen(device, 1000 + seed)
    P = H * W
    nstars = max(4, P // 30000) if nstars is None else nstars
    ncr = max(2, P // 8000) if ncr is None else ncr
    yy = torch.linspace(-1.0, 1.0, H, device=device).view(H, 1)
    xx = torch.linspace(-1.0, 1.0, W, device=device).view(1, W)
    signal = sky * (1.0 + 0.05 * xx + 0.03 * yy)
    stars = torch.zeros(H * W, device=device)
    pos = torch.randint(0, P, (nstars,), generator=g, device=device)
    amp = 10.0 ** (2.0 + 2.5 * torch.rand(nstars, generator=g, device=device))
    stars.index_add_(0, pos, amp)
    k = torch.arange(-6, 7, device=device, dtype=torch.float32)
    k1 = torch.exp(-k * k / (2 * 1.5 ** 2))
    k1 = k1 / k1.sum()
    # separable 13-tap Gaussian by shifted adds (torch's conv2d would go through MIOpen's kernel search on first use)
    st = stars.view(H, W)
    for axis in (0, 1):
        acc = torch.zeros_like(st)
        for j in range(13):
            acc += float(k1[j]) * torch.roll(st, j - 6, dims=axis)
        st = acc
    signal = signal + st.view(H, W) * (2 * 3.14159265 * 1.5 ** 2)
    frame = signal + torch.randn(H, W, generator=g, device=device) * torch.sqrt(signal + read_noise ** 2)
    truth = torch.zeros(P, dtype=torch.bool, device=device)
    cpos = torch.randint(3 * W, P - 3 * W, (ncr,), generator=g, device=device)
    camp = 300.0 + 5000.0 * torch.rand(ncr, generator=g, device=device)
    length = torch.randint(1, 4, (ncr,), generator=g, device=device)
    step = torch.where(torch.rand(ncr, generator=g, device=device) < 0.5, W + 1, W)
    flat = frame.view(-1)
    for j in range(3):
        on = length > j
        p = (cpos + j * step)[on]
        flat.index_add_(0, p, camp[on] * (0.6 ** j))
        truth[p] = True
    return frame.to(torch.float32).contiguous(), truth.view(H, W)
