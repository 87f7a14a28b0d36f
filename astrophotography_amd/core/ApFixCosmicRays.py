"""ApFixCosmicRays - L.A.Cosmic cosmic-ray cleaning (reference: core/ApFixCosmicRays.py).

Keeps the reference's API - ``ApFixCosmicRays(loglevel)``, ``process(inpdata, gain)`` -> ``(cleaned, keyword dict)``
(:242-323), ``process_file(inpfile, outfile)`` (:325-363), ``get_crdiff`` / ``get_crmask`` (:232-240),
``write_crmask_img`` / ``write_crdiff_img`` (:365-400) - and its hard-wired settings (:262-270: niter 6, readnoise 12 e-,
psffwhm 3.5, satlevel gain * 65535, gain_apply, sigclip 4.5, fsmode 'convolve').

The reference calls ``ccdproc.cosmicray_lacosmic`` (astroscrappy underneath); neither is available in the build container,
so the algorithm is restated (oracle/lacosmic_ref.py: PARITY UNPINNED) and runs as HIP kernels (csrc/lacosmic.hip).
As in ccdproc with ``gain_apply=True`` the image is multiplied by the gain, cleaned in electrons as float32, and the
reference's division by the gain brings it back to ADU.  Non-finite pixels are zeroed, masked and restored afterwards.
"""
import time
from datetime import datetime, timezone
from pathlib import Path

import numpy as np

from .. import __version__, fitsio
from . import _common


class ApFixCosmicRays:
    GOOD = 0
    AUTO_BAD = 1
    USER_BAD = 2

    def __init__(self, loglevel):
        self._name = 'ApFixCosmicRays'
        self._loglevel = loglevel
        self._logger = _common.make_logger(self._name, loglevel)
        self._imhdr = None
        self._imdata = None
        self._cleandata = None
        self._crmask = None
        self._crdiff = None
        self._cr_kw = {}

    def get_crdiff(self):
        return self._crdiff

    def get_crmask(self):
        return self._crmask

    def process_tensor(self, data_t, gain):
        """Device form: float32 / float64 tensor in ADU -> (cleaned float32 tensor in ADU, crmask uint8 tensor, settings)."""
        import torch
        from .. import ops
        la = {'gain': gain, 'sigclip': 4.5, 'readnoise': 12.0, 'psffwhm': 3.5, 'verbose': False, 'gain_apply': True,
              'satlevel': gain * 65535, 'niter': 6, 'fsmode': 'convolve'}
        self._logger.debug(f'Running cosmicray_lacosmic with the following settings: {la}')
        # data * gain in the image's own precision (ccdproc), then astroscrappy's float32 working copy
        electrons = ops.imarith(data_t.contiguous(), 'MUL', float(gain)).to(torch.float32)
        nonfinite = ~torch.isfinite(electrons)
        inmask = None
        if bool(nonfinite.any()):
            electrons = torch.where(nonfinite, torch.zeros_like(electrons), electrons)
            inmask = nonfinite.to(torch.uint8)
        clean_e, crmask, niter = ops.lacosmic(electrons, inmask=inmask, sigclip=la['sigclip'], sigfrac=0.3, objlim=5.0,
                                              readnoise=la['readnoise'], satlevel=la['satlevel'], niter=la['niter'],
                                              psffwhm=la['psffwhm'], fsmode=la['fsmode'])
        clean = ops.imarith(clean_e, 'DIV', float(gain))                 # back from electrons to ADU (:287)
        if inmask is not None:
            clean = torch.where(nonfinite, data_t.to(torch.float32), clean)
        self._logger.debug(f'L.A.Cosmic ran {niter} iteration(s).')
        return clean, crmask, la

    def process(self, inpdata, gain):
        import torch
        perf_time_start = time.perf_counter()
        self._imdata = inpdata
        self._gain = gain
        if len(tuple(inpdata.shape)) != 2 or min(inpdata.shape) < 5:
            raise ValueError(f'Error, L.A.Cosmic needs a 2-D image of at least 5 x 5 pixels, got shape {tuple(inpdata.shape)}')
        if torch.is_tensor(inpdata):
            data_t = inpdata.cuda()
            if not data_t.dtype.is_floating_point:           # integer frames are widened like the NumPy branch below
                data_t = (data_t.view(torch.int16).to(torch.int32) & 0xFFFF).double() if data_t.dtype == torch.uint16 else data_t.double()
        else:
            a = np.ascontiguousarray(inpdata)
            if a.dtype not in (np.float32, np.float64):
                a = a.astype(np.float64)
            data_t = torch.from_numpy(a).cuda()
        clean, crmask, _ = self.process_tensor(data_t, float(gain))
        self._cleandata_dev = clean
        if torch.is_tensor(inpdata):                          # results live where (and, for floats, as what) the input does
            self._cleandata = clean.to(device=inpdata.device, dtype=inpdata.dtype if inpdata.dtype.is_floating_point else clean.dtype)
            self._crmask = crmask.to(inpdata.device)
        else:
            self._cleandata = clean.cpu().numpy()
            self._crmask = crmask.cpu().numpy()
        numbad = int(crmask.sum(dtype=torch.int64))
        self._logger.info(f'{numbad} pixels in image identified as affected by cosmic rays.')
        kw_dict = {'CR_CLEAN': (True, 'Has cosmic ray removal been performed?'),
                   'CR_NPIX': (numbad, 'Number of pixels modified by lacosmic.')}
        self._crdiff = (data_t.to(self._cleandata.device) if torch.is_tensor(inpdata) else self._imdata) - self._cleandata
        self._cr_kw = kw_dict
        run_time_secs = time.perf_counter() - perf_time_start
        if numbad > 0:                                        # the reference divides by numbad unconditionally (:320)
            self._logger.debug(f'Finished CR processing in {run_time_secs:.3f} s, {1000 * run_time_secs / numbad:.3f} ms per CR pixel.')
        return self._cleandata, kw_dict

    def _update_header(self, hdr):
        kw_dict = dict(self._cr_kw)
        tnow = datetime.now().isoformat(timespec='milliseconds')
        kw_dict['CREATOR'] = (self._name, 'Software that generated this file.')
        kw_dict['DATE'] = (datetime.now(timezone.utc).isoformat(timespec='seconds'), 'UTC creation time.')
        for kw, val in kw_dict.items():
            hdr[kw] = val
        hdr['HISTORY'] = f'Processed by {self._name} {__version__} at {tnow}'

    def process_file(self, inpfile, outfile):
        self._imdata, self._imhdr, _ = _common.read_fits(self._logger, inpfile)
        gain = None
        for kw in ['GAIN', 'EGAIN']:
            if kw in self._imhdr:
                gain = float(self._imhdr[kw])
                self._logger.debug(f'Read gain value of {gain:.3f} e/ADU from {kw} keyword.')
        if gain is None:
            gain = 1.0
            self._logger.warning(f'Could not find gain value in header. Assuming gain={gain:.3f} e/ADU.')
        self.process(self._imdata, gain)
        hdr = self._imhdr.copy()
        self._update_header(hdr)
        fitsio.write(str(outfile), np.asarray(self._cleandata), hdr, overwrite=True)
        self._logger.info(f'Wrote cosmic ray cleaned image to {Path(outfile)}')

    def write_crmask_img(self, mask_file_name):
        fitsio.write(str(mask_file_name), np.asarray(self._crmask, dtype=np.uint8), None, overwrite=True)
        self._logger.info(f'Wrote cosmic ray pixel mask to {mask_file_name}')

    def write_crdiff_img(self, diff_file_name):
        fitsio.write(str(diff_file_name), np.asarray(self._crdiff), None, overwrite=True)
        self._logger.info(f'Wrote cosmic ray difference image to {diff_file_name}')
