"""ApFixBadPixels - host shell over the HIP kernel apgpu_fix_badpix_f32.

Mirrors the reference class (core/ApFixBadPixels.py): ``__init__(loglevel)`` (:28),
``fix_files(inpdata_file, badpixmask_file, outdata_file, deltapix=1)`` (:245) and the array-level
``fix_bad_pixels(data, badpixmask, deltapix=1) -> (newdata, fixed_stats)`` (:292), including the
statistics dictionary / FITS keywords BPIXNBAD, BPIX_MIN, BPIXDPIX, BPIXNREM, BPIXCORR, BPIXNFIX.
"""
import time
from datetime import datetime
from pathlib import Path

import numpy as np

from .. import __version__, fitsio
from . import _common


class ApFixBadPixels:
    """Replaces pre-identified bad pixels by the median of the surrounding good pixels."""

    MASK_GOOD = 0

    def __init__(self, loglevel):
        self._name = 'ApFixBadPixels'
        self._loglevel = loglevel
        self._logger = _common.make_logger(self._name, loglevel)
        self._min_valid = 4                      # ApFixBadPixels.py:45
        self._replace_unfixable = False          # ApFixBadPixels.py:49-50 (disabled in the reference too)
        self._logger.debug(f'{self._name} instance constructed.')

    # -- array level -----------------------------------------------------------------------------
    def fix_bad_pixels_tensor(self, data, badpixmask, deltapix=1):
        """Device form: float32 / float64 tensor [H,W] + mask tensor -> (fixed tensor, int64[3] stats tensor)."""
        from .. import ops
        return ops.fix_badpix(data, badpixmask, int(deltapix), self._min_valid)

    def fix_bad_pixels(self, data, badpixmask, deltapix=1):
        """numpy in, numpy out; same return value as the reference (ApFixBadPixels.py:292-445)."""
        import torch
        from .. import ops
        deltapix = int(deltapix)
        data = np.asarray(data)
        badpixmask = np.asarray(badpixmask)
        self._logger.info(f'fix_bad_pixels: data has {data.shape[0]} rows x {data.shape[1]} columns, dtype={data.dtype}')
        self._logger.info(f'fix_bad_pixels: mask has {badpixmask.shape[0]} rows x {badpixmask.shape[1]} columns, dtype={badpixmask.dtype}')
        if data.shape != badpixmask.shape:
            msg = (f'Error, the shape of the input data array ({data.shape})'
                   f' does not match that of the bad pixel mask array ({badpixmask.shape}).')
            self._logger.error(msg)
            raise RuntimeError(msg)
        is_float = np.issubdtype(data.dtype, np.floating)
        if not is_float:
            self._logger.warning('Pixel medians may suffer from casting truncation because'
                                 f' the input data is not a floating point datatype ({data.dtype}).')
        t0 = time.perf_counter()
        # medians in the input's own floating type (np.median): float32 stays float32, float64 stays float64; any
        # other dtype is widened to float64 (exact for integers below 2^53) and only the repaired pixels are written
        # back, so good pixels keep their bits (the reference works on data.copy(), :334)
        work_dt = np.float32 if data.dtype == np.float32 else np.float64
        d = torch.from_numpy(np.ascontiguousarray(data, dtype=work_dt)).cuda()
        bad = np.ascontiguousarray(badpixmask != ApFixBadPixels.MASK_GOOD)
        m = torch.from_numpy(bad.view(np.uint8)).cuda()
        out, st = ops.fix_badpix(d, m, deltapix, self._min_valid)
        nbad, nfixed, nnotfix = (int(x) for x in st.cpu().numpy())
        newdata = out.cpu().numpy()
        if data.dtype != work_dt:
            fixed = newdata
            newdata = data.copy()
            # the reference assigns float medians into an array of the input dtype (C truncation for integers)
            newdata[bad] = fixed[bad].astype(data.dtype) if is_float else np.trunc(fixed[bad]).astype(data.dtype)
        run_time = time.perf_counter() - t0
        npix = data.size
        pctbad = 100.0 * nbad / npix
        self._logger.debug(f'Percentage of pixels considered bad: {pctbad:.3f} ({nbad:d}/{npix:d})')
        if nbad > 0:
            self._logger.info(f'Processed {nbad} pixels in {run_time:.3f} s, {1000 * run_time / nbad:.2f} ms per bad pixel.')
        if nnotfix > 0:
            self._logger.warning(f'Could not fix {nnotfix} pixels as they had less'
                                 f' than {self._min_valid} good neighbors when deltapix={deltapix} pixels.')
        fixed_stats = {
            'numpix': (npix, 'Total number of pixels in image'),
            'BPIXNBAD': (nbad, 'Total number of bad pixels in bad pixel file'),
            'pctbad': (pctbad, 'Percentage of pixel defined bad'),
            'BPIX_MIN': (self._min_valid, 'Minimum number of good neighors needed'),
            'BPIXDPIX': (deltapix, 'Half height/width of collection region (pixels)'),
            'BPIXNREM': (nnotfix, 'Number of bad pixels not corrected'),
            'BPIXCORR': (nfixed > 0, 'True if any bad pixels were corrected'),
            'BPIXNFIX': (nfixed, 'Number of bad pixels corrected'),
        }
        return newdata, fixed_stats

    # -- file level ------------------------------------------------------------------------------
    def _write_corrected_image(self, inpdata_file, outdata_file, odata, odict):
        self._logger.debug(f'Bad pixel keywords added to output: {odict}')
        _common.check_file_exists(self._logger, inpdata_file)
        _, hdr = fitsio.read(str(inpdata_file), want_data=False)
        _common.remove_pedestal_kw(self._logger, hdr)
        for kw, val in odict.items():
            if 'BPIX' in kw:
                hdr[kw] = val
        tnow = datetime.now().isoformat(timespec='milliseconds')
        hdr['HISTORY'] = f'Applied {self._name} {__version__} at {tnow}'
        fitsio.write(str(outdata_file), odata, hdr, overwrite=True)
        self._logger.info(f'Wrote bad pixel corrected file to {outdata_file}')

    def fix_files(self, inpdata_file, badpixmask_file, outdata_file, deltapix=1):
        deltapix = int(deltapix)
        self._logger.info(f'fix_files input data file={inpdata_file}, mask file={badpixmask_file},'
                          f' output file={outdata_file}, deltapix={deltapix}')
        idata, _, _ = _common.read_fits(self._logger, inpdata_file)
        mskdata, _, _ = _common.read_fits(self._logger, badpixmask_file)
        odata, odict = self.fix_bad_pixels(idata, mskdata, deltapix)
        odict['BPIXFILE'] = (Path(badpixmask_file).name, 'Name of master bad pixel file used')
        self._write_corrected_image(inpdata_file, outdata_file, odata, odict)
