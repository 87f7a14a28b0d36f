"""ApStack / ApCombine - the N-frame stack reduction as a class (new API).

The reference has no stacking class: master darks/biases go through ``ccdproc.combine``
(scripts/ap_combine_darks.py:411-420) and registered light frames through the external SWarp program
(scripts/resample_all.sh:330-342).  ``ApStack`` exposes the per-pixel reduction both use, on slabs
resident in HBM, with the semantics of ``astropy.stats.sigma_clipped_stats(cube, axis=0)`` - the
reference's own clip function (core/ApFindBadPixels.py:191) applied along N - and optionally fused with
``ApCalibrate``'s arithmetic so raw frames are read once.
"""
from datetime import datetime, timezone
from pathlib import Path

import numpy as np

from .. import __version__, fitsio
from . import _common

METHODS = ('sigclip', 'median', 'mean')


def _apply_pedestals(slab, hdrs):
    """PEDESTAL (if present and non-zero) is ADDED to the data at read time (core/ApCalibrate.py:318-326); integer data with
    a pedestal is refused like numpy refuses `uint16 += float`."""
    import torch
    for k, h in enumerate(hdrs):
        ped = float(h['PEDESTAL']) if 'PEDESTAL' in h else 0.0
        if ped != 0:
            if not slab.dtype.is_floating_point:
                raise TypeError(f'Cannot add PEDESTAL={ped} to integer data of type {slab.dtype}.')
            slab[k] += ped
    return slab


class ApStack:
    def __init__(self, loglevel='INFO', sigma=3.0, sigma_lower=None, sigma_upper=None, maxiters=5, cenfunc='median',
                 stdfunc='std'):
        self._name = 'ApStack'
        self._logger = _common.make_logger(self._name, loglevel)
        self.sigma, self.sigma_lower, self.sigma_upper = sigma, sigma_lower, sigma_upper
        self.maxiters, self.cenfunc, self.stdfunc = maxiters, cenfunc, stdfunc

    def stack(self, frames, method='sigclip', calib=None, pixmask=None, outputs=('mean',)):
        """frames[N,H,W] device tensor (uint16|float32) -> dict of device tensors.

        method 'sigclip': clipped mean (+ 'median', 'std', 'count', 'moments' planes on request);
        'mean': plain mean (a clip with an infinite bound, one pass); 'median': np.nanmedian along N.
        calib: None or ApCalibrate.masters() + exp_ratio (+ pedestal) for the fused path."""
        from .. import ops
        from .._lib import MAX_STACK
        if method not in METHODS:
            raise ValueError(f'Error, stacking method {method} is not one of the allowed methods: {METHODS}')
        if frames.shape[0] > MAX_STACK and method != 'median':
            # beyond what one launch reduces exactly (512 frames): chunks, float64 moments added (ops.stack_sigclip_chunked)
            want = set(outputs) - {'mean', 'count', 'std'}
            if want:
                raise ValueError(f'stacks of more than {MAX_STACK} frames provide mean / count / std only, not {sorted(want)}')
            if method == 'mean':
                clip = dict(sigma=1e30, maxiters=1, cenfunc='mean', stdfunc='std')
            else:
                self._logger.warning(f'{frames.shape[0]} frames exceed the {MAX_STACK} one launch clips exactly: clipping in chunks '
                                     '(every chunk against its own statistics), moments combined in float64.')
                clip = dict(sigma=self.sigma, sigma_lower=self.sigma_lower, sigma_upper=self.sigma_upper, maxiters=self.maxiters,
                            cenfunc=self.cenfunc, stdfunc=self.stdfunc)
            return ops.stack_sigclip_chunked(frames, want_std='std' in outputs, calib=calib, pixmask=pixmask, **clip)
        if method == 'median':
            return {'median': ops.stack_median(frames, calib=calib, pixmask=pixmask)}
        if method == 'mean':
            return ops.stack_sigclip(frames, sigma=1e30, maxiters=1, cenfunc='mean', stdfunc='std', calib=calib,
                                     pixmask=pixmask, outputs=outputs)
        return ops.stack_sigclip(frames, sigma=self.sigma, sigma_lower=self.sigma_lower, sigma_upper=self.sigma_upper,
                                 maxiters=self.maxiters, cenfunc=self.cenfunc, stdfunc=self.stdfunc, calib=calib,
                                 pixmask=pixmask, outputs=outputs)

    def stack_files(self, input_files, output_file, method='sigclip', calibrator=None, extra_keywords=None):
        """Stacks FITS files into one FITS image.  With `calibrator` (an ApCalibrate) the raw frames are
        calibrated inside the stack kernel (fused, read once); otherwise they are stacked as they are."""
        import torch
        from .. import ops
        input_files = [str(f) for f in input_files]
        if not input_files:
            raise RuntimeError('No input files to stack.')
        calib = None
        if calibrator is not None:
            slab, hdrs, ratios, peds = calibrator.load_slab(input_files)
            masters = calibrator.masters()
            wide = slab.dtype == torch.float64 or any(getattr(masters[k], 'dtype', None) == torch.float64
                                                      for k in ('bias', 'dark', 'nflat'))
            if wide:
                # float64 masters (what ApMasterCal / ccdproc write, ap_combine_darks.py:437) or float64 frames: the fused
                # kernel is float32, so the frames are calibrated with NumPy's promotion first (ApCalibrate.py:439-464 in
                # float64, apgpu_calibrate_mixed) and the calibrated slab is stacked
                self._logger.warning('float64 masters / frames: calibrating in float64 before the stack (not fused); the '
                                     'calibrated values are rounded to float32 for the stack kernel.')
                slab = calibrator.calibrate_slab(slab, ratios, peds).to(torch.float32)
            else:
                calib = dict(masters, exp_ratio=ratios, pedestal=peds)
        else:
            for f in input_files:
                _common.check_file_exists(self._logger, f)
            slab, hdrs = fitsio.read_slab_device(input_files)            # pinned staging + on-device decode, one slab
            slab = _apply_pedestals(slab, hdrs)
            if slab.dtype == torch.float64:
                self._logger.warning('float64 frames are rounded to float32 for the stack kernel.')
                slab = slab.to(torch.float32)
        key = 'median' if method == 'median' else 'mean'
        res = self.stack(slab, method=method, calib=calib, outputs=('mean', 'count') if method != 'median' else ())
        out = res[key].cpu().numpy()
        hdr = hdrs[0].copy()
        for kw in ('BSCALE', 'BZERO', 'PEDESTAL'):
            if kw in hdr:
                del hdr[kw]
        hdr['NCOMBINE'] = (len(input_files), 'Number of frames combined')
        hdr['STACKMET'] = (method, 'Stack reduction method')
        if method == 'sigclip':
            hdr['STACKSIG'] = (float(self.sigma), 'Clipping threshold (sigma)')
            hdr['STACKITR'] = (-1 if self.maxiters is None else int(self.maxiters), 'Maximum clipping iterations')
            hdr['STACKCEN'] = (self.cenfunc, 'Clipping centre function')
            hdr['STACKDEV'] = (self.stdfunc, 'Clipping deviation function')
        for idx, fname in enumerate(input_files):
            hdr[f'IFILE{idx:03d}'] = Path(fname).name
        for k, v in (extra_keywords or {}).items():
            hdr[k] = v
        tnow = datetime.now().isoformat(timespec='milliseconds')
        hdr['DATE'] = (datetime.now(timezone.utc).isoformat(timespec='seconds'), 'Date/time file was created.')
        hdr['HISTORY'] = f'Processed by {self._name} {__version__} at {tnow}'
        fitsio.write(str(output_file), out, hdr, overwrite=True)
        self._logger.info(f'Wrote stacked image to {output_file}')
        return res


class ApCombine(ApStack):
    """Alias kept for the name used in BASELINE.json's north star."""

    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self._name = 'ApCombine'
