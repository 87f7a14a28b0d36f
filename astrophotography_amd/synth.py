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


def make_frames(N, masters, nflat, config_id=2, dtype=torch.float32, first_frame=0, out=None):
    """raw[N,H,W] (float32 or uint16 holding integer ADU) for the masters returned by make_masters."""
    bias, dark, scene = masters['bias'], masters['dark'], masters['scene']
    H, W = bias.shape
    device = bias.device
    if out is None:
        out = torch.empty((N, H, W), dtype=dtype, device=device)
    signal = bias + EXP_RATIO * dark + nflat * scene
    sig_noise = signal.clamp_min(0).sqrt() + 12
    for f in range(N):
        g = _gen(device, 1000 * config_id + first_frame + f)
        fr = signal + torch.randn((H, W), generator=g, device=device) * sig_noise
        cr = torch.rand((H, W), generator=g, device=device) < 0.001
        fr = fr + cr * (torch.rand((H, W), generator=g, device=device) * 4500 + 500)
        fr = fr.clamp_(0, 65535)
        if dtype == torch.float32:
            out[f] = fr
        else:
            out[f] = fr.round_().to(torch.int32).to(torch.uint16) if hasattr(torch, 'uint16') else fr
    return out
