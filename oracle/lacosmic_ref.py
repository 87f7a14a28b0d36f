"""CPU restatement of L.A.Cosmic as ApFixCosmicRays runs it (core/ApFixCosmicRays.py:267-295 ->
ccdproc.cosmicray_lacosmic -> astroscrappy.detect_cosmics).   *** TEST INFRASTRUCTURE ONLY ***   *** PARITY UNPINNED ***

ccdproc (requirements.txt:18) and astroscrappy are absent from the build container and no reference test covers the class,
so nothing here could be checked against the reference's own output.  The functions restate van Dokkum's (2001) algorithm in
the structure of astroscrappy's detect_cosmics (float32 planes, separable median filters with copied borders, 'meanmask'
cleaning, 'convolve' fine-structure mode with a Gaussian kernel) as far as its published source is remembered; border
conventions (zero padding of the convolution and the dilations, neighbours outside the image dropped by the Laplacian) are
this build's choice.  The HIP kernels (csrc/lacosmic.hip) are tested against THIS file bit for bit.
"""
import numpy as np

F = np.float32


def sepmedfilt(a, size):
    """Median of `size` along rows, then along columns; pixels closer than size // 2 to a border are copied."""
    a = np.asarray(a, F)
    H, W = a.shape
    h = size // 2
    r = a.copy()
    if W >= size:
        win = np.lib.stride_tricks.sliding_window_view(a, size, axis=1)
        r[:, h:W - h] = np.median(win, axis=-1)
    out = r.copy()
    if H >= size:
        win = np.lib.stride_tricks.sliding_window_view(r, size, axis=0)
        out[h:H - h, :] = np.median(win, axis=-1)
    return out


def laplace_rebin(a):
    """rebin(clip0(laplace(subsample2(a)))): kernel 0 -1 0 / -1 4 -1 / 0 -1 0 on the 2x subsampled image (a neighbour outside
    the image is dropped), negatives clipped, 2x2 block average.  Per pixel v with neighbours u, d, l, r the sub-pixels are
    4v - u - l - v - v etc., evaluated left to right in float32."""
    a = np.asarray(a, F)
    H, W = a.shape
    z = F(0)
    u = np.zeros_like(a); u[1:] = a[:-1]
    d = np.zeros_like(a); d[:-1] = a[1:]
    l = np.zeros_like(a); l[:, 1:] = a[:, :-1]
    r = np.zeros_like(a); r[:, :-1] = a[:, 1:]
    four = F(4) * a

    def sub(p, q):
        return np.maximum((((four - p) - q) - a) - a, z)
    tl, tr, bl, br = sub(u, l), sub(u, r), sub(d, l), sub(d, r)
    return (((tl + tr) + bl) + br) * F(0.25)


def gausskernel(fwhm, size=7):
    x = np.tile(np.arange(size) - size // 2, (size, 1)).astype(np.float64)
    y = x.T.copy()
    sigma2 = fwhm * fwhm / 2.35482 / 2.35482
    k = np.exp(-0.5 * (x * x + y * y) / sigma2).astype(F)
    return (k / k.sum()).astype(F)


def convolve7(a, k):
    """7 x 7 correlation with zero padding, accumulated in row-major kernel order in float32."""
    a = np.asarray(a, F)
    H, W = a.shape
    pad = np.zeros((H + 6, W + 6), F)
    pad[3:H + 3, 3:W + 3] = a
    valid = np.zeros((H + 6, W + 6), bool)
    valid[3:H + 3, 3:W + 3] = True
    acc = np.zeros((H, W), F)
    for dy in range(7):
        for dx in range(7):
            term = k[dy, dx] * pad[dy:dy + H, dx:dx + W]
            acc = np.where(valid[dy:dy + H, dx:dx + W], acc + term, acc)
    return acc


def dilate(m, shape):
    """Binary dilation, zero outside: 3 = 3x3 square, 5 = 5x5 without its corners."""
    m = np.asarray(m, bool)
    H, W = m.shape
    R = shape // 2
    pad = np.zeros((H + 2 * R, W + 2 * R), bool)
    pad[R:R + H, R:R + W] = m
    out = np.zeros((H, W), bool)
    for dy in range(-R, R + 1):
        for dx in range(-R, R + 1):
            if shape == 5 and abs(dy) == 2 and abs(dx) == 2:
                continue
            out |= pad[R + dy:R + dy + H, R + dx:R + dx + W]
    return out


def satmask(data, inmask, satlevel):
    """astroscrappy update_mask: cores of saturated stars (>= satlevel where the 7-median is above satlevel / 10) dilated
    twice by the 5x5 kernel, OR the input mask grown by one pixel."""
    data = np.asarray(data, F)
    sat = (data >= F(satlevel)) & (sepmedfilt(data, 7) > F(satlevel) / F(10.0))
    sat = dilate(dilate(sat, 5), 5)
    if inmask is not None:
        sat |= dilate(np.asarray(inmask) != 0, 3)
    return sat


def iterate(clean, mask, crmask, sigclip, sigfrac, objlim, readnoise, psfk, background):
    """One detect_cosmics iteration, in place on clean / crmask; returns the number of cosmic-ray pixels found."""
    s = laplace_rebin(clean)
    m5 = np.maximum(sepmedfilt(clean, 7), F(0.00001))
    noise = np.sqrt(m5 + F(readnoise) * F(readnoise))
    s = s / (F(2.0) * noise)
    sp = s - sepmedfilt(s, 7)
    f = convolve7(clean, psfk) if psfk is not None else sepmedfilt(clean, 5)
    f = np.maximum((f - sepmedfilt(f, 9)) / noise, F(0.01))
    good = ~mask
    cr = good & (sp > F(sigclip)) & ((sp / f) > F(objlim))
    cr = dilate(cr, 3) & good & (sp > F(sigclip))
    cr = dilate(cr, 3) & good & (sp > F(sigfrac) * F(sigclip))
    n = int(cr.sum())
    crmask |= cr
    H, W = clean.shape
    bad = crmask | mask
    src = clean.copy()
    for (r, c) in zip(*np.nonzero(crmask)):
        if r < 2 or r >= H - 2 or c < 2 or c >= W - 2:
            continue
        tot, cnt = F(0), 0
        for dy in range(-2, 3):
            for dx in range(-2, 3):
                if not bad[r + dy, c + dx]:
                    tot = F(tot + src[r + dy, c + dx])
                    cnt += 1
        clean[r, c] = F(tot / F(cnt)) if cnt > 0 else F(background)
    return n


def detect_cosmics(data, gain=1.0, sigclip=4.5, sigfrac=0.3, objlim=5.0, readnoise=12.0, satlevel=65535.0, niter=6, psffwhm=3.5,
                   fsmode='convolve', inmask=None):
    """ccdproc.cosmicray_lacosmic(data, gain_apply=True, ...) as ApFixCosmicRays calls it: (cleaned image in ELECTRONS, float32;
    crmask bool).  Non-finite pixels are zeroed and masked (the kernels' contract; astroscrappy would propagate NaN)."""
    clean = (np.asarray(data) * gain).astype(F)
    nonfinite = ~np.isfinite(clean)
    clean[nonfinite] = F(0)
    base = nonfinite if inmask is None else (nonfinite | (np.asarray(inmask) != 0))
    mask = satmask(clean, base, satlevel)
    goodvals = clean[~mask]
    background = F(np.median(goodvals)) if goodvals.size else F(0)
    psfk = gausskernel(psffwhm, 7) if fsmode == 'convolve' else None
    crmask = np.zeros(clean.shape, bool)
    for _ in range(niter):
        if iterate(clean, mask, crmask, sigclip, sigfrac, objlim, readnoise, psfk, background) == 0:
            break
    return clean, crmask
