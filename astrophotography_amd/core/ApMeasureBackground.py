"""ApMeasureBackground - large-scale sky-background model (reference: core/ApMeasureBackground.py).

Keeps the reference's API - ``ApMeasureBackground(loglevel)``, ``process_data(imdata, imhdr, nbg_rows, nbg_cols,
min_bgheight, min_bgwidth, bg_filter_width, bg_badbox_pctile, bg_sigmaclip)`` (:331-415), ``process_files`` (:417-471),
``get_bgimage`` (:473-477), ``write_bgimage`` (:479-518) - over HIP kernels (csrc/background.hip).

The reference hands the numerics to photutils (:154-157 the source mask, :404-410 ``Background2D(boxsize, filter_size,
mask, exclude_percentile, SigmaClip(nsigma), MedianBackground)``).  photutils is not available in the build container,
so its published algorithms are restated (oracle/background_ref.py: PARITY UNPINNED) and implemented as:

  device  global sigma-clipped statistics (A3 kernels) -> detection threshold -> threshold mask (A4 kernel)
          -> 8-connected components >= 5 pixels, dilated 13 x 13 (apgpu_source_mask_u8)
          -> per-box sigma-clipped median / std over every unmasked pixel (apgpu_box_clipped_stats_f32)
  host    the ny x nx mesh (a few hundred numbers): boxes with too many masked / clipped pixels are excluded and filled
          by inverse-distance weighting, 3 x 3 median filter, cubic B-spline prefilter
  device  the spline evaluated at every pixel (apgpu_spline_zoom_f64 = scipy.ndimage.zoom order 3, 'reflect', grid mode)
"""
import logging
from datetime import datetime
from pathlib import Path

import numpy as np

from .. import __version__, fitsio
from . import _common


def _fill_excluded(mesh, good, n_neighbors=10, power=1.0):
    """Background2D._interpolate_meshes: inverse-distance weighting over the k nearest good boxes."""
    ny, nx = mesh.shape
    gy, gx = np.nonzero(good)
    vals = mesh[good]
    out = mesh.copy()                       # a good box is its own nearest neighbour at distance 0: it keeps its value
    k = min(n_neighbors, len(vals))
    for y, x in zip(*np.nonzero(~good)):
        d = np.hypot(gy - y, gx - x)
        order = np.argsort(d, kind='stable')[:k]
        w = 1.0 / d[order] ** power
        out[y, x] = np.sum(w * vals[order]) / np.sum(w)
    return out


def _nanmedian_filter(mesh, size):
    """Background2D._filter_meshes: nanmedian over a size x size window, NaN beyond the mesh."""
    if size <= 1:
        return mesh
    h = size // 2
    ny, nx = mesh.shape
    pad = np.full((ny + 2 * h, nx + 2 * h), np.nan)
    pad[h:h + ny, h:h + nx] = mesh
    win = np.stack([pad[dy:dy + ny, dx:dx + nx] for dy in range(size) for dx in range(size)], 0)
    return np.nanmedian(win, axis=0)


def _bspline3_prefilter(mesh):
    """Cubic B-spline coefficients of the mesh: scipy.ndimage.spline_filter(mesh, order=3, mode='reflect'), i.e. the
    recursive filter with pole z = sqrt(3) - 2 and scipy's half-sample-symmetric initial conditions, along both axes
    (reproduces scipy to 1e-15 for every mesh size, including its slightly inexact start-up for meshes of < 8 boxes)."""
    z = np.sqrt(3.0) - 2.0

    def filter_axis0(f):
        n = f.shape[0]
        c = f * 6.0                                         # gain (1 - z)(1 - 1/z)
        zn = z ** n
        c0 = c[0].copy()
        acc = c[0] + zn * c[n - 1]
        zi = z
        for i in range(1, n):
            mirror = acc if i == n - 1 else c[n - 1 - i]    # scipy accumulates in place: the last term sees the running c[0]
            acc = acc + zi * (c[i] + zn * mirror)
            zi *= z
        c[0] = acc * (z / (1.0 - zn * zn)) + c0
        for i in range(1, n):
            c[i] = c[i] + z * c[i - 1]
        c[n - 1] = c[n - 1] * (z / (z - 1.0))
        for i in range(n - 2, -1, -1):
            c[i] = z * (c[i + 1] - c[i])
        return c

    c = filter_axis0(np.array(mesh, dtype=np.float64))
    c = filter_axis0(np.ascontiguousarray(c.T)).T
    return np.ascontiguousarray(c)


class ApMeasureBackground:
    def __init__(self, loglevel):
        self._name = 'ApMeasureBackground'
        self._loglevel = loglevel
        # defaults of the reference (ApMeasureBackground.py:80-91)
        self._default_minwdth = 48
        self._default_minhght = 48
        self._default_nbgcols = 16
        self._default_nbgrows = 16
        self._default_filtsiz = 3
        self._default_pctlile = 25.0
        self._default_nsigma = 3.0
        self._boxsize = (self._default_minhght, self._default_minwdth)
        self._filtersize = (self._default_filtsiz, self._default_filtsiz)
        self._exclude_pctile = self._default_pctlile
        self._nsigma = self._default_nsigma
        self._imdata = None
        self._imhdr = None
        self._bgdata = None
        self._bgdata_dev = None
        self._logger = _common.make_logger(self._name, loglevel)

    # -- source mask (:142-175) ---------------------------------------------------------------------
    def _make_source_mask(self, data_t):
        """detect_threshold(nsigma=2, SigmaClip(3, maxiters=10)) -> detect_sources(npixels=5) -> make_source_mask(13):
        device tensor uint8 [H, W]."""
        import torch
        from .. import ops
        st = ops.sigclip_global(data_t, sigma=3.0, maxiters=10)          # float32 statistics as numpy computes them
        # threshold = mean + 2 * std in float32 (photutils: background + error * nsigma on float32 arrays); formed on the
        # device from the 10-number statistics vector: no host round trip
        thr = (st[0].float() + st[2].float() * 2.0).double()
        th = torch.stack([torch.full_like(thr, -float('inf')), thr]).contiguous()
        above, _ = ops.threshold_mask(data_t, thresholds=th)
        mask, nsrc = ops.source_mask(above, min_pixels=5, dilate_size=13)
        if self._logger.isEnabledFor(logging.DEBUG):        # the two counts cost a reduction and two host round trips
            npix = mask.numel()
            npos = int(mask.sum(dtype=torch.int64))
            self._logger.debug(f'Source mask: {int(nsrc.item())} sources, {npos} of {npix} pixels masked ({100.0 * npos / npix:.2f} percent).')
        return mask

    # -- box size (:251-329) -------------------------------------------------------------------------
    def _set_bgbox_size(self, imrows, imcols, nbg_rows, nbg_cols, min_bgheight, min_bgwidth):
        actual_nbg_rows = self._default_nbgrows if nbg_rows is None else nbg_rows
        actual_nbg_cols = self._default_nbgcols if nbg_cols is None else nbg_cols
        actual_min_bgheight = self._default_minhght if min_bgheight is None else min_bgheight
        # the reference assigns min_bgheight here when min_bgwidth is given (:295-296); kept, so that a drop-in run
        # produces the same mesh geometry
        actual_min_bgwidth = self._default_minwdth if min_bgwidth is None else min_bgheight
        if actual_min_bgwidth is None:
            actual_min_bgwidth = self._default_minwdth
        self._logger.debug(f'Background box size constraints: {actual_nbg_rows} rows x {actual_nbg_cols} columns of background boxes.'
                           f' Minimum box size {actual_min_bgheight} pixels high x {actual_min_bgwidth} pixels wide.')
        quantum = 2
        box_height = max(actual_min_bgheight, quantum * (1 + int(imrows / (quantum * actual_nbg_rows))))
        box_width = max(actual_min_bgwidth, quantum * (1 + int(imcols / (quantum * actual_nbg_cols))))
        nrows, ncols = actual_nbg_rows * box_height, actual_nbg_cols * box_width
        if nrows < imrows:
            if (imrows - nrows) > quantum * actual_nbg_rows:
                self._logger.error(f'Error in _set_bgbox_size logic: missing {imrows - nrows} rows using box height of {box_height} rows.')
            else:
                box_height += quantum
        if ncols < imcols:
            if (imcols - ncols) > quantum * actual_nbg_cols:
                self._logger.error(f'Error in _set_bgbox_size logic: missing {imcols - ncols} cols using box width of {box_width} cols.')
            else:
                box_width += quantum
        self._logger.info(f'Background box size is {box_height} pixels high x {box_width} pixels wide.')
        self._box_height, self._box_width = box_height, box_width
        self._boxsize = (box_height, box_width)

    # -- reference API -------------------------------------------------------------------------------
    def process_data(self, imdata, imhdr=None, nbg_rows=None, nbg_cols=None, min_bgheight=None, min_bgwidth=None,
                     bg_filter_width=None, bg_badbox_pctile=None, bg_sigmaclip=None):
        import torch
        from .. import ops
        self._imdata = imdata
        self._imhdr = imhdr
        if torch.is_tensor(imdata):
            data_t = imdata.to(device='cuda', dtype=torch.float32).contiguous()
        else:
            data_t = torch.from_numpy(np.ascontiguousarray(imdata, dtype=np.float32)).cuda()
        if data_t.dim() != 2:
            raise RuntimeError(f'Expected a 2-D image, got shape {tuple(data_t.shape)}.')
        H, W = data_t.shape
        srcmask = self._make_source_mask(data_t)
        if bg_filter_width is not None:
            self._filtersize = (bg_filter_width, bg_filter_width)
        if bg_badbox_pctile is not None:
            self._exclude_pctile = bg_badbox_pctile
        if bg_sigmaclip is not None:
            self._nsigma = bg_sigmaclip
        self._set_bgbox_size(H, W, nbg_rows, nbg_cols, min_bgheight, min_bgwidth)
        bh, bw = self._boxsize
        self._logger.debug('Background estimator settings used: estimator=MedianBackground'
                           f', clipper=SigmaClip with sigma={self._nsigma}, boxsize={self._boxsize}, filter_size={self._filtersize}'
                           f', exclude_percentile={self._exclude_pctile}')
        stats = ops.box_clipped_stats(data_t, srcmask, bh, bw, sigma=float(self._nsigma), maxiters=5).cpu().numpy()
        med, std, nfin = stats[..., 0], stats[..., 1], stats[..., 2]
        npix = bh * bw
        # a box is excluded when more than exclude_percentile percent of its pixels are masked - input mask, the
        # padding of the last row / column of boxes, non-finite pixels and the pixels the sigma clip rejected
        good = ((npix - nfin) <= self._exclude_pctile / 100.0 * npix) & (nfin > 0)
        if not good.any():
            raise ValueError(f'All boxes contain > {self._exclude_pctile / 100.0 * npix} ({self._exclude_pctile} percent per box) masked '
                             'pixels (or all are completely masked). Please check your data or increase "exclude_percentile" '
                             'to allow more boxes to be included.')
        mesh = np.where(good, med, np.nan)
        rms = np.where(good, std, np.nan)
        if not good.all():
            self._logger.debug(f'{int((~good).sum())} of {good.size} background boxes excluded; filled from their neighbours.')
            mesh, rms = _fill_excluded(mesh, good), _fill_excluded(rms, good)
        mesh = _nanmedian_filter(mesh, int(self._filtersize[0]))
        rms = _nanmedian_filter(rms, int(self._filtersize[0]))
        self._mesh, self._rms_mesh, self._mesh_good = mesh, rms, good
        if np.ptp(mesh) == 0:
            bg = torch.full((H, W), float(mesh.min()), dtype=torch.float64, device=data_t.device)
        else:
            coef = torch.from_numpy(_bspline3_prefilter(mesh)).to(data_t.device)
            bg = ops.spline_zoom(coef, bh, bw, H, W, float(mesh.min()), float(mesh.max()))
        self._bgdata_dev = bg
        self._bgdata = None                     # host copy on demand (get_bgimage): 8 bytes per pixel over PCIe
        self._bgmedian = float(np.median(mesh))
        self._bgmedian_rms = float(np.median(rms))
        self._logger.info(f'Estimated median background level: {self._bgmedian:.3f}+/-{self._bgmedian_rms:.3f}')
        self._logger.debug(f'Background mesh shape: {mesh.shape}')

    def process_files(self, input_fits, srclist_fits=None, nbg_rows=None, nbg_cols=None, min_bgheight=None, min_bgwidth=None,
                      bg_filter_width=None, bg_badbox_pctile=None, bg_sigmaclip=None):
        data, hdr, _ = _common.read_fits(self._logger, input_fits)
        if srclist_fits is not None:
            self._logger.warning('Source list processing not yet implemented.')
        self.process_data(data, hdr, nbg_rows, nbg_cols, min_bgheight, min_bgwidth, bg_filter_width, bg_badbox_pctile, bg_sigmaclip)

    def get_bgimage(self):
        if self._bgdata is None and self._bgdata_dev is not None:
            self._bgdata = self._bgdata_dev.cpu().numpy()
        return self._bgdata

    def get_bgimage_device(self):
        """The background image as the float64 device tensor the zoom kernel wrote (no host copy)."""
        return self._bgdata_dev

    def write_bgimage(self, output_bgfits):
        if self._bgdata_dev is None:
            raise RuntimeError('Error, you can not write a background image before generating one using process_data or process_files.')
        hdr = self._imhdr.copy() if self._imhdr is not None else fitsio.Header()
        _common.remove_pedestal_kw(self._logger, hdr)
        tnow = datetime.now().isoformat(timespec='milliseconds')
        hdr['HISTORY'] = f'Applied {self._name} {__version__} at {tnow}'
        hdr['IMAGETYP'] = 'Background Sky'
        fitsio.write_device(str(output_bgfits), self._bgdata_dev, hdr, overwrite=True)
        self._logger.info(f'Wrote estimated background data to {Path(output_bgfits)}')
