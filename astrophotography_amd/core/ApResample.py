"""ApResample - registration resample + co-add of calibrated frames (new API).

The reference does this step outside Python: ``scripts/resample_all.sh`` collects the navigated files of one
filter, derives a flux scale ``FSCALE = 1/EXPOSURE`` per file (:283-312; 1.0 in SUM mode), and runs SWarp with
``RESAMPLING_TYPE LANCZOS3``, ``OVERSAMPLING 4``, ``COMBINE_TYPE MEDIAN | WEIGHTED | SUM``, ``GAIN_KEYWORD EGAIN`` and a
weight-map output (:60-73, 108-118, 330-342).  ``ApResample`` keeps that contract on the GPU for frames whose registration is a 2x3 affine
transform per frame (output pixel -> input pixel): Lanczos-3 resampling onto the common grid
(``ops.resample_affine``), the flux scaling, the combine, and a weight image (= number of frames that
contributed to each pixel).  Files that carry a TAN WCS (astrometry.net) are registered through the sky with one
affine per 16 x 64 output tile (wcs.tile_affines); distortion polynomials (SIP / TPV) are not supported.
"""
from datetime import datetime, timezone
from pathlib import Path

import numpy as np

from .. import __version__, fitsio
from . import _common

COMBINE_TYPES = ('MEDIAN', 'AVERAGE', 'WEIGHTED', 'SUM', 'CLIPPED')


class ApResample:
    def __init__(self, loglevel='INFO', combine='MEDIAN', sigma=3.0, maxiters=5, n_phases=1024, conserve_flux=True, oversampling=1,
                 gain_keyword='EGAIN'):
        self._name = 'ApResample'
        self._logger = _common.make_logger(self._name, loglevel)
        combine = str(combine).upper()
        if combine not in COMBINE_TYPES:
            raise ValueError(f'Error, combine type {combine} is not one of the allowed types: {COMBINE_TYPES}')
        self.combine, self.sigma, self.maxiters, self.n_phases = combine, sigma, maxiters, n_phases
        self.conserve_flux = bool(conserve_flux)            # FSCALASTRO_TYPE VARIABLE (resample_all.sh:129)
        self.oversampling = int(oversampling)               # OVERSAMPLING (resample_all.sh:112: 4)
        self.gain_keyword = gain_keyword                    # GAIN_KEYWORD (resample_all.sh:111: EGAIN)
        if not 1 <= self.oversampling <= 16:
            raise ValueError(f'Error, oversampling {oversampling} is not in 1..16')

    def coadd(self, frames, affines, fscale=None, mask=None, out_shape=None, weights=None, fine_affines=None, fused=False):
        """frames [N,H,W] float32 device tensor -> dict(image, count[, weight]) device tensors.  fused: ops.coadd's one-launch
        form (CLIPPED / AVERAGE of up to 16 frames, no resampled slab in memory)."""
        from .. import ops
        return ops.coadd(frames, affines, fscale=fscale, mask=mask, out_shape=out_shape, combine=self.combine,
                         sigma=self.sigma, maxiters=self.maxiters, n_phases=self.n_phases, conserve_flux=self.conserve_flux,
                         oversampling=self.oversampling, weights=weights, fine_affines=fine_affines, fused=fused)

    def _output_gain(self, gains, fscale, weights):
        """Effective gain of the co-add in electrons per output unit, written as GAIN when every input carries the
        GAIN_KEYWORD (resample_all.sh:111 passes EGAIN to SWarp for this purpose).  This build's definition, by propagating
        the Poisson variance of frames of equal signal (SWarp's own formula is not reproduced): a frame with gain g_i scaled by
        f_i has gain G_i = g_i / f_i; SUM: 1 / G = sum 1 / G_i; a mean with weights w_i (1 for AVERAGE, CLIPPED and - as an
        approximation - MEDIAN): G = (sum w_i)^2 / sum(w_i^2 / G_i).  None without the keyword."""
        if any(g is None for g in gains):
            return None
        g = np.asarray(gains, dtype=np.float64) / np.asarray(fscale, dtype=np.float64)
        if self.combine == 'SUM':
            return float(1.0 / np.sum(1.0 / g))
        w = np.ones(len(g)) if weights is None or self.combine != 'WEIGHTED' else np.asarray(weights, dtype=np.float64)
        return float(np.sum(w) ** 2 / np.sum(w * w / g))

    def _exposure(self, hdr, fname):
        for kw in ('EXPOSURE', 'EXPTIME'):                   # same order as resample_all.sh:283-297
            if kw in hdr:
                return float(hdr[kw])
        raise RuntimeError(f'Error, could not find EXPOSURE keyword in {fname}.')

    def coadd_files(self, input_files, affines, output_file, weight_file=None, mask_file=None, out_shape=None,
                    center=None, pixscale=None):
        """Resamples and combines FITS files.  `affines`: one [a00, a01, a02, a10, a11, a12] per file, or None to
        register through the files' celestial WCS (RA---TAN / DEC--TAN headers, as astrometry.net writes them):
        the output grid is then north-up TAN at `center` = (ra, dec) degrees with `pixscale` arcsec per pixel and
        `out_shape` pixels - SWarp's -CENTER / -PIXEL_SCALE / -IMAGE_SIZE (resample_all.sh:334-338) - or, without
        `center`, the first file's own WCS and shape.
        Flux scale per file = 1 / EXPOSURE (or EXPTIME), except 1.0 for SUM (resample_all.sh:298, 305-309)."""
        import torch
        input_files = [str(f) for f in input_files]
        if not input_files:
            raise RuntimeError('No input files to resample.')
        use_wcs = affines is None
        if not use_wcs:
            affines = np.asarray(affines, dtype=np.float64).reshape(-1, 6)
            if len(affines) != len(input_files):
                raise RuntimeError(f'Error, {len(affines)} transforms given for {len(input_files)} files.')
        from .ApStack import _apply_pedestals
        from .. import fitsio
        for f in input_files:
            _common.check_file_exists(self._logger, f)
        # one float32 slab in HBM, filled through pinned staging + on-device decode (fitsio.read_slab_device)
        slab, hdrs = fitsio.read_slab_device(input_files, dtype=torch.float32)
        slab = _apply_pedestals(slab, hdrs)
        fscale, texp, gains = [], 0.0, []
        for f, hdr in zip(input_files, hdrs):
            exp = self._exposure(hdr, f)
            texp += exp
            gains.append(float(hdr[self.gain_keyword]) if self.gain_keyword and self.gain_keyword in hdr else None)
            fscale.append(1.0 if self.combine == 'SUM' else 1.0 / exp)
            self._logger.info(f'  File {Path(f).name:40s} EXPOSURE {exp:8.3f} FSCALE {fscale[-1]:8.6f}')
        in_shape = tuple(slab.shape[1:])
        mask = None
        if mask_file is not None:
            m, _, _ = _common.read_fits(self._logger, mask_file)
            if m.shape != in_shape:
                raise RuntimeError(f'Error, mask shape {m.shape} differs from the image shape {in_shape}.')
            mask = torch.from_numpy((np.asarray(m) != 0).astype(np.uint8)).cuda()
        out_wcs = None
        if use_wcs:
            from .. import wcs as apwcs
            in_wcs = [apwcs.TanWcs.from_header(h) for h in hdrs]
            if center is not None:
                if pixscale is None or out_shape is None:
                    raise RuntimeError('Error, center needs pixscale and out_shape as well.')
                out_wcs = apwcs.TanWcs.from_center(float(center[0]), float(center[1]), float(pixscale), out_shape)
            else:
                out_wcs = in_wcs[0]
                out_shape = out_shape or in_shape
            fine_affines = None
            if self.oversampling > 1:
                n = self.oversampling
                fine_wcs, fine_shape = out_wcs.oversampled(n), (out_shape[0] * n, out_shape[1] * n)
                fine_affines = np.stack([apwcs.tile_affines(fine_wcs, w, fine_shape, tile_scale=n) for w in in_wcs], 0)
                affines = np.zeros((len(in_wcs), 6))
            else:
                affines = np.stack([apwcs.tile_affines(out_wcs, w, out_shape) for w in in_wcs], 0)
            self._logger.info(f'Registered {len(in_wcs)} files through their TAN WCS onto a {out_shape[1]}x{out_shape[0]} grid.')
        else:
            fine_affines = None
        weights = None
        if self.combine == 'WEIGHTED':
            from .. import ops
            weights = ops.background_weights(slab, np.asarray(fscale, np.float64))
            for f, wgt in zip(input_files, weights):
                self._logger.info(f'  File {Path(f).name:40s} WEIGHT {wgt:12.6g}')
        res = self.coadd(slab, affines, fscale=np.asarray(fscale, np.float32), mask=mask, out_shape=out_shape, weights=weights,
                         fine_affines=fine_affines)
        hdr = hdrs[0].copy()
        for kw in ('BSCALE', 'BZERO', 'PEDESTAL'):
            if kw in hdr:
                del hdr[kw]
        hdr['NCOMBINE'] = (len(input_files), 'Number of frames combined')
        hdr['COMBINET'] = (self.combine, 'Co-add combine type')
        hdr['RESAMPT'] = ('LANCZOS3', 'Resampling kernel')
        hdr['OVERSAMP'] = (self.oversampling, 'Sub-samples per output pixel and axis')
        gain_out = self._output_gain(gains, fscale, weights)
        if gain_out is not None:
            hdr['GAIN'] = (gain_out, f'Effective gain from {self.gain_keyword} of the inputs')
        if out_wcs is not None:
            for k in ('CD1_1', 'CD1_2', 'CD2_1', 'CD2_2', 'CDELT1', 'CDELT2', 'CROTA1', 'CROTA2', 'PC1_1', 'PC1_2', 'PC2_1', 'PC2_2'):
                if k in hdr:
                    del hdr[k]
            for k, v in out_wcs.header_cards().items():
                hdr[k] = v
        hdr['TEXPTIME'] = (texp, '[s] Total exposure of the inputs')
        if self.combine != 'SUM':
            hdr['BUNIT'] = ('adu/s', 'Pixel value units (flux scaled by 1/EXPOSURE)')
        for idx, fname in enumerate(input_files):
            hdr[f'IFILE{idx:03d}'] = Path(fname).name
        tnow = datetime.now().isoformat(timespec='milliseconds')
        hdr['DATE'] = (datetime.now(timezone.utc).isoformat(timespec='seconds'), 'Date/time file was created.')
        hdr['HISTORY'] = f'Processed by {self._name} {__version__} at {tnow}'
        fitsio.write(str(output_file), res['image'].cpu().numpy(), hdr, overwrite=True)
        self._logger.info(f'Wrote co-added image to {output_file}')
        if weight_file is not None:
            wh = fitsio.Header()
            wh['NCOMBINE'] = (len(input_files), 'Number of frames combined')
            wimg = res['weight'] if 'weight' in res else res['count']
            what = 'sum of the contributing frames\' weights' if 'weight' in res else 'frames contributing per pixel'
            wh['HISTORY'] = f'Weight map ({what}) by {self._name} {__version__} at {tnow}'
            fitsio.write(str(weight_file), wimg.cpu().numpy().astype(np.float32), wh, overwrite=True)
            self._logger.info(f'Wrote weight image to {weight_file}')
        return res
