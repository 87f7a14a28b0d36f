"""CPU restatement of the sky-background mesh of ApMeasureBackground (core/ApMeasureBackground.py:142-175, 382-415).
*** TEST INFRASTRUCTURE ONLY ***   *** PARITY UNPINNED ***

The reference calls photutils (detect_threshold, detect_sources, SegmentationImage.make_source_mask, Background2D with
MedianBackground + SigmaClip, BkgZoomInterpolator).  photutils is absent from the build container (requirements.txt:20
pins photutils>=1.10) and no reference test covers this class, so nothing here could be checked against the reference's
own output: the functions restate photutils' published algorithms with NumPy / SciPy (scipy.ndimage supplies label,
binary_dilation, generic_filter and zoom - the very routines photutils calls) and the pinned sigma-clip of oracle/apref.c.
The HIP kernels are tested against THIS file; agreement with photutils itself is unverified.
"""
import numpy as np
from scipy import ndimage

from oracle import apref


def detect_threshold(data, nsigma=2.0):
    """photutils.segmentation.detect_threshold(data, nsigma, sigma_clip=SigmaClip(sigma=3, maxiters=10)): sigma-clipped
    mean + nsigma * sigma-clipped std of the whole image, in the image's own float32 arithmetic (numpy nan-functions)."""
    st = apref.sigclip_global(np.ascontiguousarray(data, np.float32), sigma=3.0, maxiters=10)
    mean, std = np.float32(st['mean']), np.float32(st['std'])
    return np.float32(mean + np.float32(std * np.float32(nsigma)))


def make_source_mask(data, nsigma=2.0, npixels=5, dilate_size=13):
    """ApMeasureBackground._make_source_mask (:154-157): detect_sources(data, threshold, npixels) with 8-connectivity, then
    SegmentationImage.make_source_mask(size): binary dilation with a size x size square footprint."""
    thr = detect_threshold(data, nsigma)
    above = np.asarray(data, np.float32) > thr
    lab, nlab = ndimage.label(above, structure=np.ones((3, 3), int))
    sizes = np.bincount(lab.ravel(), minlength=nlab + 1)
    keep = sizes >= npixels
    keep[0] = False
    seg = keep[lab]
    mask = ndimage.binary_dilation(seg, structure=np.ones((dilate_size, dilate_size), bool))
    return mask, int(keep.sum()), thr


def box_clipped_stats(data, mask, box_h, box_w, sigma=3.0, maxiters=5):
    """Per-box SigmaClip(sigma, maxiters) + nanmedian / nanstd (Background2D with MedianBackground; edge_method 'pad':
    the image is padded with masked pixels to whole boxes).  Returns (median, std, nfinal, nmasked0) as [ny, nx] arrays."""
    H, W = data.shape
    ny, nx = -(-H // box_h), -(-W // box_w)
    pad = np.full((ny * box_h, nx * box_w), np.nan, np.float32)
    pad[:H, :W] = np.where(np.isfinite(data), data, np.nan)
    if mask is not None:
        m = np.zeros(pad.shape, bool)
        m[:H, :W] = np.asarray(mask) != 0
        pad[m] = np.nan
    boxes = pad.reshape(ny, box_h, nx, box_w).transpose(1, 3, 0, 2).reshape(box_h * box_w, ny, nx)
    r = apref.stack_sigclip(np.ascontiguousarray(boxes), sigma=sigma, maxiters=maxiters, want=('median', 'std', 'count'))
    nmasked0 = np.isnan(boxes).sum(0)
    return r['median'], r['std'], r['count'].astype(np.int64), nmasked0.astype(np.int64)


def fill_excluded(mesh, good, n_neighbors=10, power=1.0):
    """Background2D._interpolate_meshes: ShepardIDWInterpolator over the good boxes (k nearest, weights 1 / d^power),
    evaluated at every box; a good box keeps its own value (distance 0)."""
    ny, nx = mesh.shape
    gy, gx = np.nonzero(good)
    vals = mesh[good]
    out = np.empty_like(mesh)
    k = min(n_neighbors, len(vals))
    for y in range(ny):
        for x in range(nx):
            d = np.hypot(gy - y, gx - x)
            order = np.argsort(d, kind='stable')[:k]
            dk = d[order]
            if dk[0] == 0.0:
                out[y, x] = vals[order[0]]
            else:
                w = 1.0 / dk ** power
                out[y, x] = np.sum(w * vals[order]) / np.sum(w)
    return out


def median_filter_mesh(mesh, size=3):
    """Background2D._filter_meshes: nanmedian over a size x size window, NaN outside the mesh."""
    if size <= 1:
        return mesh
    return ndimage.generic_filter(mesh, np.nanmedian, size=size, mode='constant', cval=np.nan)


def zoom_mesh(mesh, box_h, box_w, H, W):
    """BkgZoomInterpolator: scipy.ndimage.zoom(order=3, mode='reflect', grid_mode=True), cropped, clipped to the mesh range."""
    if np.ptp(mesh) == 0:
        return np.zeros((H, W)) + mesh.min()
    z = ndimage.zoom(mesh, (box_h, box_w), order=3, mode='reflect', grid_mode=True)[:H, :W]
    return np.clip(z, mesh.min(), mesh.max())


def background2d(data, mask, box_h, box_w, filter_size=3, exclude_percentile=25.0, sigma=3.0, maxiters=5):
    """Background2D(data, (box_h, box_w), filter_size, mask, exclude_percentile, SigmaClip(sigma), MedianBackground):
    dict(background [H, W] float64, mesh, rms_mesh, background_median, background_rms_median, good)."""
    H, W = data.shape
    med, std, nfin, _ = box_clipped_stats(data, mask, box_h, box_w, sigma, maxiters)
    npix = box_h * box_w
    good = (npix - nfin) <= exclude_percentile / 100.0 * npix      # masked: input mask, padding, non-finite AND clipped pixels
    good &= nfin > 0
    if not good.any():
        raise ValueError('All boxes contain > %s (%s percent per box) masked pixels (or all are completely masked). '
                         'Please check your data or increase "exclude_percentile" to allow more boxes to be included.'
                         % (exclude_percentile / 100.0 * npix, exclude_percentile))
    mesh = np.where(good, med, np.nan)
    rms = np.where(good, std, np.nan)
    if not good.all():
        mesh = fill_excluded(mesh, good)
        rms = fill_excluded(rms, good)
    mesh = median_filter_mesh(mesh, filter_size)
    rms = median_filter_mesh(rms, filter_size)
    return dict(background=zoom_mesh(mesh, box_h, box_w, H, W), mesh=mesh, rms_mesh=rms, good=good,
                background_median=float(np.median(mesh)), background_rms_median=float(np.median(rms)))
