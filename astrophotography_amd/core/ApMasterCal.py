"""ApMasterCal - master dark / bias / flat creation (reference: scripts/ap_combine_darks.py:100-441).

The reference delegates file discovery to ``ccdproc.ImageFileCollection`` and all arithmetic to
``ccdproc.combine(method='average', sigma_clip=True, low=high=5, sigma_clip_func=np.ma.median,
sigma_clip_dev_func=mad_std, mem_limit=5e8)`` (:394-420).  Here the files are read once into an [N,H,W]
slab in HBM and reduced by ONE launch of the stack kernel configured the same way (one clipping pass,
median centre, 1.4826*MAD deviation, 5 sigma): no 500 MB tiling and no re-reading of every file per tile.

The product has ccdproc's CCDData layout (ap_combine_darks.py:437 ``master.write``): a float64 primary HDU holding
the kernel's float64 mean (``mean_f64``: never narrowed to float32 on the way), a MASK extension (pixels with every
input rejected) and an UNCERT extension (float64 std of the survivors / sqrt(their number)).  ApCalibrate then
calibrates with these float64 masters in float64, as the reference does.  ccdproc itself is not available in the build
container to pin the arithmetic (parity unpinned).
"""
import fnmatch
import os
from datetime import datetime, timezone
from pathlib import Path

import numpy as np

from .. import fitsio
from . import _common

_FITS_EXT = ('.fit', '.fits', '.fts')


class ApMasterCal:
    # Which published Combiner.sigma_clipping make_master follows: 'astropy' (ccdproc >= 2.2 delegates to
    # astropy.stats.sigma_clip - what requirements.txt:18 `ccdproc>=2.1.0` resolves to today) or 'legacy' (ccdproc <= 2.1's own
    # loop).  They differ for columns holding a non-finite value (unclipped in the astropy form) and, on float64 frames, within an
    # ulp of a bound (golden group G12 holds both, run for real).  A class attribute: the constructor keeps the
    # reference's signature (scripts/ap_combine_darks.py:112-116).
    ccdproc_form = 'astropy'

    def __init__(self, rootdir, exclude_pattern, telescop, temptol, loglevel):
        self._logger = _common.make_logger('ApMasterCal', loglevel)
        self._loglevel = loglevel
        self._rootdir = rootdir
        self._telescop = telescop
        self._temptol = float(temptol)
        self._summary_kw = ['file', 'date-obs', 'telescop', 'imagetyp', 'filter', 'exptime', 'set-temp', 'ccd-temp',
                            'naxis1', 'naxis2']
        self._data_dir = Path(rootdir)
        self._set_temperature = None
        self._summary = self._create_file_collection(self._data_dir, exclude_pattern, None)
        self._file_list = self._check_files(True)
        self._summary = self._create_file_collection(self._data_dir, exclude_pattern, self._file_list)
        self._logger.debug('ApMasterCal constructor completed.')

    # -- file collection (ccdproc.ImageFileCollection stand-in: header summary of FITS files) ------
    def _create_file_collection(self, data_dir, exclude_pattern, file_list):
        if not Path(data_dir).is_dir():
            raise RuntimeError(f'Cannot find {data_dir}. Not a valid path or file.')
        if file_list is not None:
            names = list(file_list)
            self._logger.info(f'Looking for FITS files in {data_dir}, including only files in the list: {file_list}')
        else:
            self._logger.info(f'Looking for FITS files in {data_dir}, excluding files matching the pattern "{exclude_pattern}"')
            names = sorted(n for n in os.listdir(data_dir)
                           if n.lower().endswith(_FITS_EXT) and not (exclude_pattern and fnmatch.fnmatch(n, exclude_pattern)))
        rows = []
        for n in names:
            hdr = fitsio.getheader(str(Path(data_dir) / n))
            row = {'file': n}
            for kw in self._summary_kw[1:]:
                row[kw] = hdr.get(kw.upper(), '')
            rows.append(row)
        self._logger.info(f'Found {len(rows)} FITS files matching the constraints.')
        return rows

    def _values(self, kw, unique=False):
        vals = [r[kw] for r in self._summary]
        if unique:
            out = []
            for v in vals:
                if v not in out:
                    out.append(v)
            return out
        return vals

    def _check_files(self, list_all=None):
        """Type / size / exposure / set-temp must be unique (else RuntimeError); files whose CCD-TEMP is
        further than temptol from the set temperature are dropped (ap_combine_darks.py:150-287)."""
        raw_file_list = self._values('file')
        if not raw_file_list:
            raise RuntimeError(f'No FITS files found in {self._data_dir}.')
        uniq = {kw: self._values(kw, unique=True) for kw in self._summary_kw}
        if list_all:
            for kw in self._summary_kw[1:]:
                self._logger.debug(f'For keyword {kw} there are {len(uniq[kw])} values: {uniq[kw]}')
        for kw in ['telescop', 'imagetyp', 'naxis1', 'naxis2', 'exptime', 'set-temp']:
            if len(uniq[kw]) > 1:
                msg = f'Error, there are {len(uniq[kw])} unique values of {kw} in the files being processed: {uniq[kw]}'
                self._logger.error(msg)
                raise RuntimeError(msg)
        self._imgtype = uniq['imagetyp'][0]
        self._exptime = uniq['exptime'][0]
        telescop = str(uniq['telescop'][0])
        if not telescop.strip():
            self._logger.warning(f'TELESCOP keyword empty or missing in input files. Using {self._telescop} instead.')
        else:
            self._telescop = telescop.strip()
        set_temperature = None
        val = uniq['set-temp'][0]
        if isinstance(val, str):
            if not val.strip():
                self._logger.warning('No numeric value found for SET-TEMP. Will use median of CCD-TEMP instead.')
        else:
            set_temperature = float(val)
        temps = self._values('ccd-temp')
        if len(uniq['ccd-temp']) == 1 and isinstance(uniq['ccd-temp'][0], str) and not uniq['ccd-temp'][0].strip():
            self._logger.warning('No files contain CCD-TEMP metadata. Continuing assuming all files obtained at the same temperature.')
            return raw_file_list
        if set_temperature is None:
            set_temperature = float(np.median([float(t) for t in temps]))
        temp_min, temp_max = set_temperature - self._temptol, set_temperature + self._temptol
        self._logger.info(f'Selecting only files with CCD-TEMP between {temp_min:.2f} and {temp_max:.2f} degrees C.')
        self._set_temperature = set_temperature
        good = []
        for fname, temp in zip(raw_file_list, temps):
            if temp_min <= float(temp) <= temp_max:
                good.append(fname)
            else:
                self._logger.warning(f'Excluding {fname} as CCD-TEMP={float(temp):.2f} outside allowed range.')
        self._logger.info(f'Updated file list contains {len(good)} files ({len(raw_file_list)} before filtering).')
        return good

    def _generate_final_keywords(self):
        creation_datestr = datetime.now(timezone.utc).isoformat(timespec='seconds')
        raw_imgtype = str(self._imgtype).lower()
        if 'bias' in raw_imgtype:
            imgtype = 'MASTER BIAS'
        elif 'dark' in raw_imgtype:
            imgtype = 'MASTER DARK'
        elif 'flat' in raw_imgtype:
            imgtype = 'MASTER FLAT'
        else:
            self._logger.warning(f'Unexpected input image type: {raw_imgtype}')
            imgtype = raw_imgtype
        kw = {'IMAGETYP': (imgtype, 'Type of file'), 'TELESCOP': (self._telescop, 'Telescope used.'),
              'CREATOR': ('ApMasterCal', 'Software that generated this file.')}
        if self._set_temperature is not None:
            kw['SET-TEMP'] = (self._set_temperature, '[Celsius] Desired CCD temperature')
            kw['CCD-TEMP'] = kw['SET-TEMP']
        kw['DATE'] = (creation_datestr, 'Date/time file was created.')
        for idx, fname in enumerate(self._values('file')):
            kw[f'IFILE{idx:03d}'] = fname
        return kw

    def make_master(self, output_master_file):
        import torch
        from .. import ops
        kw_dict = self._generate_final_keywords()
        files = [self._data_dir / n for n in self._values('file')]
        self._logger.debug(f'About to combine {len(files)} {kw_dict["IMAGETYP"][0]} files, method=average sigma_clip=True '
                           'sig_clip_lothresh=5 sig_clip_hithresh=5 (median / mad_std, one pass).')
        # one slab in HBM, filled file by file through pinned staging + on-device decode (fitsio.read_slab_device)
        slab, hdrs = fitsio.read_slab_device([str(f) for f in files])
        if slab.dtype == torch.float64:
            # float64 frames (BITPIX -64): ccdproc combines in float64 anyway; here the float64 combine kernel
            res = ops.combine_f64(slab, sigma_lower=5.0, sigma_upper=5.0, maxiters=1, cenfunc='median', stdfunc='mad_std',
                                  form=self.ccdproc_form)
        else:
            res = None
        if res is None:
            res = ops.stack_sigclip(slab, sigma=5.0, maxiters=1, cenfunc='median', stdfunc='mad_std',
                                    outputs=('mean_f64', 'count', 'std_f64'),
                                    nonfinite_unclipped=(self.ccdproc_form == 'astropy'))
        # ccdproc's CCDData product (ap_combine_darks.py:411-439): float64 primary, MASK = pixels with every
        # input rejected, UNCERT = std of the surviving values / sqrt(their number) - all in float64
        master = res['mean_f64'].cpu().numpy()
        count = res['count'].cpu().numpy()
        all_masked = (count == 0).astype(np.uint8)
        with np.errstate(divide='ignore', invalid='ignore'):
            uncert = res['std_f64'].cpu().numpy() / np.sqrt(count.astype(np.float64))
        res['mean'] = res['mean_f64']
        hdr = hdrs[0].copy()
        for k in ('BSCALE', 'BZERO', 'UT', 'TIME-OBS', 'SWOWNER', 'SWCREATE', 'SBSTDVER'):
            if k in hdr:
                del hdr[k]
        hdr['NCOMBINE'] = len(files)
        hdr['COMBINED'] = True
        hdr['BUNIT'] = 'adu'
        for k, v in kw_dict.items():
            hdr[k] = v
        fitsio.write(str(output_master_file), master, hdr, overwrite=True,
                     extensions=[('MASK', all_masked, None), ('UNCERT', uncert, {'UTYPE': ('StdDevUncertainty', '')})])
        self._logger.info(f'Wrote combined calibration file: {output_master_file}')
        return res
