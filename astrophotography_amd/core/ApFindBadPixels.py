"""ApFindBadPixels - host shell over apgpu_sigclip_global_f32 / apgpu_threshold_mask_f32 /
apgpu_mask_add_rects_u8 (reference: core/ApFindBadPixels.py).

``ApFindBadPixels(darkfile, sigma, loglevel)`` does all the work in the constructor (:30-68):
global sigma-clipped statistics of the dark (astropy sigma_clipped_stats defaults: maxiters=5, median
centre, std; :191), thresholds median -/+ sigma*std (:194-195), uint8 mask of pixels strictly outside
(:199-209).  ``add_user_badpix(yaml)`` adds USER_BAD (=2) over user columns/rows/rectangles (1-based,
inclusive; :70-158, 414-438), ``get_mask()`` returns the mask, ``write_mask(file)`` writes it with the
reference's keywords (:371-412, 445-474).
"""
from datetime import datetime, timezone
from pathlib import Path

import numpy as np

from .. import __version__, fitsio
from . import _common


class ApFindBadPixels:
    GOOD = 0
    AUTO_BAD = 1
    USER_BAD = 2

    def __init__(self, darkfile, sigma, loglevel):
        self._name = 'ApFindBadPixels'
        self._loglevel = loglevel
        self._logger = _common.make_logger(self._name, loglevel)
        self._imfile = darkfile
        self._imextnum = 0
        self._userfile = None
        self._nbad_auto = 0
        self._nbad_user = 0
        # the reference's CLI forgets type=float for --sigma (ap_find_badpix.py:53-58) and then fails with
        # a TypeError for any user value; a numeric string is accepted here
        self._sigma = float(sigma)
        self._imdata, self._imhdr, _ = _common.read_fits(self._logger, darkfile)
        self._mask_host = None
        self._generate_sigmaclip_mask(self._imdata, self._sigma)

    # -------------------------------------------------------------------------------------------
    def _generate_sigmaclip_mask(self, data, sigma):
        import torch
        from .. import ops
        npix = data.size
        self._logger.debug(f'Generating a bad pixel mask using sigma={sigma} clipping on the input image data values.')
        # float32 darks: numpy's float32 statistics; integer / float64 darks: float64 statistics (numpy's
        # dtype rules), both reproduced exactly on the device
        if data.dtype == np.float32:
            d = torch.from_numpy(np.ascontiguousarray(data)).cuda()
            stats = ops.sigclip_global(d, sigma=sigma, maxiters=5)
            cast = np.float32
        else:
            if data.dtype == np.uint16:
                dsrc = ops.to_device_u16(data)
            else:
                dsrc = torch.from_numpy(np.ascontiguousarray(data)).cuda()
            stats = ops.sigclip_global(dsrc, sigma=sigma, maxiters=5)
            cast = np.float64
            # the threshold comparison runs in float32 (numpy promotes uint16 vs a float scalar to float32)
            d = torch.from_numpy(np.ascontiguousarray(data, dtype=np.float32)).cuda()
        # One device pipeline, no host read in between: statistics -> thresholds -> mask.  ApFindBadPixels.py:194-195 forms
        # median -/+ sigma * std from np.float32 scalars and a python float, i.e. in float64 (numpy 1.x promotion); the
        # statistics vector holds the float32 (or float64) statistics exactly, so the same float64 expression is
        # evaluated on the device and handed to the mask kernel as a tensor.
        thr = torch.stack((stats[1] - sigma * stats[2], stats[1] + sigma * stats[2]))
        mask, nbad = ops.threshold_mask(d, thresholds=thr)
        self._badpixmask_dev = mask
        self._mask_host = None
        # the only synchronisation: everything the log lines and get_stats() report, in one read
        s = torch.cat((stats, thr, nbad.to(torch.float64))).cpu().numpy()
        mean, med, std = cast(s[0]), cast(s[1]), cast(s[2])
        lothresh, hithresh = float(s[10]), float(s[11])
        self._logger.debug(f'Sigma-clipped mean={mean:.2f}, median={med:.2f}, and madstddev={std:.2f} values (ADU).')
        self._logger.info(f'Good pixels have values between {lothresh:.2f} and {hithresh:.2f} ADU.')
        self._stats = dict(mean=mean, median=med, std=std, lothresh=lothresh, hithresh=hithresh, niter=int(s[5]))
        nbad = int(s[12])
        self._logger.info(f'Out of {npix} pixels, {nbad} are bad ({100 * (nbad / npix):.4f}%).')
        self._nbad_auto = nbad

    def _rects_from_user(self, badcols, badrows, badrect):
        """1-based inclusive user entries -> 0-based half-open [r0, r1, c0, c1] + touched-pixel count."""
        nrows, ncols = self._imdata.shape
        rects = []
        num_user_bad = 0
        for col in badcols or []:
            col1 = col - 1
            if col1 < 0 or col1 >= ncols:
                self._logger.warning(f'Warning, column {col} (1-based) outside image.')
                continue
            rects.append([0, nrows, col1, col])
            num_user_bad += nrows
        for row in badrows or []:
            row1 = row - 1
            if row1 < 0 or row1 >= nrows:
                self._logger.warning(f'Warning, row {row} (1-based) outside image.')
                continue
            rects.append([row1, row, 0, ncols])
            num_user_bad += ncols
        for rect in badrect or []:
            if len(rect) != 4:
                self._logger.warning(f'Error, expecting 4-element list, got {rect}. Skipping.')
                continue
            row1, row2, col1, col2 = rect[0] - 1, rect[1], rect[2] - 1, rect[3]
            if row1 < 0 or row2 > nrows:
                self._logger.warning(f'Warning, row range {row1}:{row2} (0-based) outside image.')
            elif col1 < 0 or col2 > ncols:
                self._logger.warning(f'Warning, column range {col1}:{col2} (0-based) outside image.')
            else:
                rects.append([row1, row2, col1, col2])
                num_user_bad += (row2 - row1) * (col2 - col1)
        return rects, num_user_bad

    def _read_user_badpix(self, user_badpix_file):
        import yaml
        user_badpix_file = _common.check_file_exists(self._logger, user_badpix_file)
        with open(user_badpix_file) as bpfile:
            yobj = yaml.safe_load(bpfile.read()) or {}
        out = []
        for key in ('bad_columns', 'bad_rows', 'bad_rectangles'):
            val = yobj.get(key)          # a missing key is "none" here (the reference raises TypeError)
            out.append(list(val) if val else None)
        return out

    def add_user_badpix(self, user_badpix_file):
        from .. import ops
        user_badpix_file = Path(user_badpix_file).expanduser()
        self._logger.info(f'Processing user-defined bad pixels from {user_badpix_file}')
        badcols, badrows, badrect = self._read_user_badpix(user_badpix_file)
        self._userfile = user_badpix_file
        rects, num_user_bad = self._rects_from_user(badcols, badrows, badrect)
        if rects:
            ops.mask_add_rects(self._badpixmask_dev, rects, ApFindBadPixels.USER_BAD)
            self._mask_host = None
        self._nbad_user = num_user_bad
        self._logger.debug(f'Total number of user-defined bad pixels applied to mask: {num_user_bad}')

    def get_mask(self):
        """The bad pixel mask as a uint8 numpy array (values are sums of AUTO_BAD / USER_BAD flags)."""
        if self._mask_host is None:
            self._mask_host = self._badpixmask_dev.cpu().numpy()
        return self._mask_host

    def get_mask_tensor(self):
        return self._badpixmask_dev

    def get_stats(self):
        return dict(self._stats)

    def _update_header(self, hdr):
        copy_list = ['TELESCOP', 'INSTRUME', 'SET-TEMP', 'CCD-TEMP', 'XPIXSZ', 'YPIXSZ', 'XBINNING', 'YBINNING',
                     'XORGSUBF', 'YORGSUBF', 'SITELAT', 'SITELONG']
        tnow = datetime.now().isoformat(timespec='milliseconds')
        creation_datestr = datetime.now(timezone.utc).isoformat(timespec='seconds')
        hdr['IMAGETYP'] = ('BADPIX', 'Type of file')
        hdr['CREATOR'] = (self._name, 'Software that generated this file.')
        hdr['DATE'] = (creation_datestr, 'UTC creation time.')
        hdr['DATAFILE'] = (str(self._imfile), 'Data file used to identify bad pixels.')
        if self._userfile is not None:
            hdr['USERFILE'] = (self._userfile.name, 'User-defined bad pixel file.')
        hdr['NBADAUTO'] = (int(self._nbad_auto), 'Number of algorithm-detected bad pixels.')
        hdr['NBADUSER'] = (int(self._nbad_user), 'Number of user-defined bad pixels.')
        for kw in copy_list:
            if kw in self._imhdr:
                hdr[kw] = (self._imhdr[kw], self._imhdr.comment(kw))
        hdr['HISTORY'] = f'Processed by {self._name} {__version__} at {tnow}'

    def write_mask(self, mask_file_name):
        hdr = fitsio.Header()
        hdr['EXTEND'] = True
        self._update_header(hdr)
        fitsio.write(str(mask_file_name), self.get_mask(), hdr, overwrite=True)
        self._logger.info(f'Wrote bad pixel mask to {mask_file_name}')
