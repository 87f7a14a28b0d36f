"""ApCalibrate - host shell over the HIP calibration kernels (reference: core/ApCalibrate.py).

Keeps the reference's file-based API - ``ApCalibrate(master_bias_file, master_dark_file, master_flat_file,
master_badpix_file, loglevel, dark_still_biased=None)`` (:48-113) and ``calibrate(raw_image, cal_image,
delta_pix, norm_flat, fixcosmic)`` (:406-509) - and adds what the reference lists as a TODO ("Allow
calibration of images in memory", :3): slab entry points that keep the masters resident in HBM and
calibrate (and optionally stack) N frames per kernel launch.

Arithmetic (ApCalibrate.py:439-464), each operation rounded separately:
    x = raw - bias;  D = dark - bias if dark_still_biased else dark;  x = x - exp_ratio * D
    y = where(nflat != 0, x / nflat, x)   with nflat = flat / nanmean(flat)   (:166-190)
in float32 when every array is float32 / uint16, and with NumPy's per-operation promotion to float64 as soon as a
float64 array takes part: _read_fits converts only non-float FITS data to float32 (:301-305), and the masters
ApMasterCal writes are float64 (scripts/ap_combine_darks.py:437) - the calibrated file is then BITPIX -64.
"""
import time
from datetime import datetime
from pathlib import Path

import numpy as np

from .. import __version__, fitsio
from . import _common
from .ApFixBadPixels import ApFixBadPixels


class ApCalibrate:
    MEAN_FULL = 0       #: Mean value of entire flat field.
    MEDIAN_FULL = 1     #: Median value of entire flat field (not implemented in the reference either).

    def __init__(self, master_bias_file, master_dark_file, master_flat_file, master_badpix_file, loglevel,
                 dark_still_biased=None):
        self._name = 'ApCalibrate'
        self._version = __version__
        self._master_bias_file = master_bias_file
        self._master_dark_file = master_dark_file
        self._master_flat_file = master_flat_file
        self._master_badpix_file = master_badpix_file
        self._loglevel = loglevel
        self._logger = _common.make_logger(self._name, loglevel)
        self._master_bias = Path(master_bias_file)
        self._master_dark = Path(master_dark_file)
        self._dark_still_biased = bool(dark_still_biased) if dark_still_biased is not None else False
        self._bpix = None
        self._crfix = None
        self._bias_data, self._bias_hdr = self._read_master(self._master_bias)
        self._dark_data, self._dark_hdr = self._read_master(self._master_dark)
        if self._bias_data.shape != self._dark_data.shape:
            raise RuntimeError(f'Master bias {tuple(self._bias_data.shape)} and master dark '
                               f'{tuple(self._dark_data.shape)} have different shapes.')
        self._norm_flat = None
        if master_flat_file is not None:
            self._master_flat = Path(master_flat_file)
            self._flat_method = ApCalibrate.MEAN_FULL
            self._logger.info(f'Reading master flat field {self._master_flat.name}')
            flat_data, _ = self._read_master(self._master_flat)
            self._norm_flat = self._generate_flat(flat_data, self._flat_method)
        if master_badpix_file is not None:
            self._bpix = ApFixBadPixels(loglevel)
            self._master_bpix = Path(master_badpix_file)
            msk, self._mskhdr, _ = _common.read_fits(self._logger, self._master_bpix)
            import torch
            self._mskdata = torch.from_numpy(np.ascontiguousarray(msk != 0).view(np.uint8)).cuda()

    # -------------------------------------------------------------------------------------------
    @staticmethod
    def _to_f32(t):
        """ApCalibrate._read_fits (:301-305): floating-point data stays as stored, everything else becomes float32."""
        import torch
        if t.dtype in (torch.float32, torch.float64):
            return t
        if t.dtype == torch.uint16:                       # exact widening
            return (t.view(torch.int16).to(torch.int32) & 0xFFFF).to(torch.float32)
        return t.to(torch.float32)

    def _read_master(self, path):
        """Master frame -> float32 / float64 device tensor (+ header); payload decoded on the device
        (fitsio.read_device).  PEDESTAL handling as _read_fits (core/ApCalibrate.py:318-326): a non-zero pedestal is added."""
        import torch
        from .. import ops
        path = _common.check_file_exists(self._logger, path)
        self._logger.info('Loading extension {} of FITS file {}'.format(0, path))
        data, hdr = fitsio.read_device(str(path))
        if hdr['NAXIS'] == 3:
            self._logger.error('Error, 3-D handling has not been implemented yet.')
            raise SystemExit(1)
        if data is None or data.dim() != 2:
            raise RuntimeError(f'{path}: expected a 2-D primary image, found NAXIS={hdr["NAXIS"]}.')
        data = self._to_f32(data).contiguous()
        if 'PEDESTAL' in hdr and float(hdr['PEDESTAL']) != 0:
            self._logger.debug(f'Removing a PEDESTAL value of {float(hdr["PEDESTAL"])} ADU.')
            data = ops.imarith(data, 'ADD', float(hdr['PEDESTAL']))
        return data, hdr

    def _find_exptime_ratio(self, img_hdr, dark_hdr):
        """Image / dark exposure-time ratio; EXPOSURE wins over EXPTIME (ApCalibrate.py:128-164, same messages)."""
        def exposure(hdr):
            return next((float(hdr[kw]) for kw in ('EXPOSURE', 'EXPTIME') if kw in hdr), None)

        times = (exposure(img_hdr), exposure(dark_hdr))
        problems = {(True, True): 'Could not determine exposure time for both image and dark.',
                    (True, False): 'Could not determine exposure time for image (dark exposure found).',
                    (False, True): 'Could not determine exposure time for dark (img exposure found).'}
        msg = problems.get((times[0] is None, times[1] is None))
        if msg is not None:
            self._logger.error(msg)
            raise RuntimeError(msg)
        exp_ratio = times[0] / times[1]
        self._logger.info(f'Image to dark exposure time ratio: {exp_ratio:.3f}')
        return exp_ratio

    def _get_gain(self, hdr):
        """Gain in e/ADU from GAIN, then EGAIN (the later keyword wins, as in the reference's loop); 1.0 if neither is there
        (ApCalibrate.py:192-208)."""
        gain = None
        for kw in ['GAIN', 'EGAIN']:
            if kw in hdr:
                gain = float(hdr[kw])
                self._logger.debug(f'Read gain value of {gain:.3f} e/ADU from {kw} keyword.')
        if gain is None:
            gain = 1.0
            self._logger.warning(f'Could not find gain value in header. Assuming gain={gain:.3f} e/ADU.')
        return gain

    def _generate_flat(self, flat_data, flat_method):
        from .. import ops
        if flat_method != ApCalibrate.MEAN_FULL:
            msg = f'Error, flat field normalization method {flat_method} has not been implemented yet.'
            self._logger.error(msg)
            raise RuntimeError(msg)
        out_flat, norm = ops.flat_normalize(flat_data)
        self._norm_factor = float(norm.item())
        self._logger.info(f'Flat field normalization factor: {self._norm_factor:.2f}')
        return out_flat

    def _read_raw(self, raw_image):
        """Raw light frame -> (device tensor uint16|float32|float64, header, pedestal to add on the device)."""
        import torch
        raw_image = _common.check_file_exists(self._logger, raw_image)
        t, hdr = fitsio.read_device(str(raw_image))
        if hdr['NAXIS'] == 3:
            self._logger.error('Error, 3-D handling has not been implemented yet.')
            raise SystemExit(1)
        if t is None or t.dim() != 2:
            raise RuntimeError(f'{raw_image}: expected a 2-D primary image, found NAXIS={hdr["NAXIS"]}.')
        pedestal = float(hdr['PEDESTAL']) if 'PEDESTAL' in hdr else 0.0
        if t.dtype not in (torch.uint16, torch.float32, torch.float64):
            t = t.to(torch.float32)                          # ApCalibrate.py:304-307 for integers
        if tuple(t.shape) != tuple(self._bias_data.shape):
            raise RuntimeError(f'{raw_image.name} has shape {tuple(t.shape)}, the masters have {tuple(self._bias_data.shape)}.')
        return t, hdr, pedestal

    def _write_corrected_image(self, inpdata_file, outdata_file, odata, odict):
        """Copy of the raw header minus PEDESTAL/BSCALE/BZERO, + keywords, + HISTORY (ApCalibrate.py:348-404)."""
        self._logger.debug(f'FITS header keywords added to output: {odict}')
        _common.check_file_exists(self._logger, inpdata_file)
        _, hdr = fitsio.read(str(inpdata_file), want_data=False)
        _common.remove_pedestal_kw(self._logger, hdr)
        for kw in ['BSCALE', 'BZERO']:
            if kw in hdr:
                del hdr[kw]
        for kw, val in odict.items():
            hdr[kw] = val
        tnow = datetime.now().isoformat(timespec='milliseconds')
        hdr['HISTORY'] = f'Processed by {self._name} {self._version} at {tnow}'
        if isinstance(odata, np.ndarray):
            fitsio.write(str(outdata_file), odata, hdr, overwrite=True)
        else:
            # big-endian encode on the device; calibrate_files hands a pool of writer threads over (self._write_pool)
            fitsio.write_device(str(outdata_file), odata, hdr, overwrite=True, pool=getattr(self, '_write_pool', None))
        self._logger.info(f'Wrote bias/dark/flat corrected file to {outdata_file}')

    def _base_keywords(self):
        odict = {'BIASCORR': (True, 'True if bias subtracted.'),
                 'BIASFILE': (self._master_bias.name, 'Master bias file used.'),
                 'DARKCORR': (True, 'True if scaled dark subtracted.'),
                 'DARKFILE': (self._master_dark.name, 'Master dark file used.'),
                 'BUNIT': ('adu', 'Pixel value units.')}
        if self._norm_flat is not None:
            odict['FLATCORR'] = (True, 'True if flat field applied.')
            odict['FLATFILE'] = (self._master_flat.name, 'Master flat file used.')
        return odict

    # -- reference API ------------------------------------------------------------------------------
    def calibrate(self, raw_image, cal_image, delta_pix, norm_flat, fixcosmic):
        from .. import ops
        perf_time_start = time.perf_counter()
        raw_image = Path(raw_image)
        raw, raw_hdr, pedestal = self._read_raw(raw_image)
        if self._dark_still_biased:
            self._logger.info('Subtracting bias from dark')
        else:
            self._logger.debug('Dark assumed to already be bias-subtracted.')
        exp_ratio = self._find_exptime_ratio(raw_hdr, self._dark_hdr)
        img_bdf = ops.calibrate(raw, self._bias_data, self._dark_data, self._norm_flat, exp_ratio,
                                pedestal=pedestal if pedestal != 0 else None,
                                dark_still_biased=self._dark_still_biased)
        odict = self._base_keywords()
        if self._norm_flat is not None:
            if norm_flat is not None:
                self._logger.debug(f'Writing normalized flat field to {norm_flat}')
                self._write_corrected_image(raw_image, norm_flat, self._norm_flat, {})
        else:
            self._logger.info('No flat field correction applied.')
        if self._bpix is not None:
            fixed, st = ops.fix_badpix(img_bdf, self._mskdata, int(delta_pix), self._bpix._min_valid)
            nbad, nfixed, nnotfix = (int(x) for x in st.cpu().numpy())
            img_bdf = fixed
            odict['BPIXFILE'] = (self._master_bpix.name, 'Name of master bad pixel file used')
            odict['BPIXNBAD'] = (nbad, 'Total number of bad pixels in bad pixel file')
            odict['BPIX_MIN'] = (self._bpix._min_valid, 'Minimum number of good neighors needed')
            odict['BPIXDPIX'] = (int(delta_pix), 'Half height/width of collection region (pixels)')
            odict['BPIXNREM'] = (nnotfix, 'Number of bad pixels not corrected')
            odict['BPIXCORR'] = (nfixed > 0, 'True if any bad pixels were corrected')
            odict['BPIXNFIX'] = (nfixed, 'Number of bad pixels corrected')
        else:
            self._logger.info('No bad pixel correction applied.')
        if fixcosmic:                                           # ApCalibrate.py:490-497
            self._logger.info('Correcting cosmic rays...')
            gain = self._get_gain(raw_hdr)
            if self._crfix is None:
                from .ApFixCosmicRays import ApFixCosmicRays
                self._crfix = ApFixCosmicRays(self._loglevel)
            img_clean, crmask, _ = self._crfix.process_tensor(img_bdf, gain)
            numbad = int(crmask.sum())
            self._logger.info(f'{numbad} pixels in image identified as affected by cosmic rays.')
            odict['CR_CLEAN'] = (True, 'Has cosmic ray removal been performed?')
            odict['CR_NPIX'] = (numbad, 'Number of pixels modified by lacosmic.')
            img_bdf = img_clean
        run_time_secs = time.perf_counter() - perf_time_start
        self._logger.info(f'Writing calibrated image to {cal_image}')
        self._write_corrected_image(raw_image, cal_image, img_bdf, odict)
        self._logger.info(f'Calibrated {raw_image.name} in {run_time_secs:.3f} seconds.')

    # -- slab API (new) -----------------------------------------------------------------------------
    def masters(self):
        """The resident masters: dict(bias, dark, nflat, dark_still_biased) of device tensors."""
        return dict(bias=self._bias_data, dark=self._dark_data, nflat=self._norm_flat,
                    dark_still_biased=self._dark_still_biased)

    def load_slab(self, raw_images):
        """Reads N raw frames into one contiguous [N,H,W] device slab (fitsio.read_slab_device: pinned double-buffered
        staging, payload decoded on the device straight into the slab).

        Returns (slab, headers, exp_ratio[N], pedestal[N] or None)."""
        import torch
        files = [str(_common.check_file_exists(self._logger, f)) for f in raw_images]
        hdrs0 = [fitsio.read(f, want_data=False)[1] for f in files]
        wide = [int(h['BITPIX']) == -64 for h in hdrs0]
        if any(wide) and not all(wide):
            raise TypeError('raw frames mix float64 with other types: their results have different dtypes, calibrate '
                            'them one at a time')
        for f, h in zip(files, hdrs0):
            if h['NAXIS'] == 3:
                self._logger.error('Error, 3-D handling has not been implemented yet.')
                raise SystemExit(1)
        slab, hdrs = fitsio.read_slab_device(files)          # uint16 if all are, else float32 (exact widening) / float64
        if tuple(slab.shape[1:]) != tuple(self._bias_data.shape):
            raise RuntimeError(f'{Path(files[0]).name} has shape {tuple(slab.shape[1:])}, the masters have {tuple(self._bias_data.shape)}.')
        ratios = [self._find_exptime_ratio(h, self._dark_hdr) for h in hdrs]
        peds = [float(h['PEDESTAL']) if 'PEDESTAL' in h else 0.0 for h in hdrs]
        return slab, hdrs, ratios, (peds if any(p != 0 for p in peds) else None)

    def calibrate_slab(self, slab, exp_ratio, pedestal=None):
        """raw[N,H,W] (uint16|float32|float64 device tensor) -> calibrated [N,H,W] in one launch (float32, or float64 if
        the frames or any master are float64)."""
        from .. import ops
        return ops.calibrate(slab, self._bias_data, self._dark_data, self._norm_flat, exp_ratio, pedestal=pedestal,
                             dark_still_biased=self._dark_still_biased)

    def calibrate_files(self, raw_images, cal_images, delta_pix=2, fixcosmic=False, timings=None):
        """Batch form of calibrate(): one slab read, one calibrate launch, N outputs - what a long-lived process should call
        instead of one ap_calibrate.py process per frame (scripts/calibrate_all.sh:406-411 starts a Python interpreter per frame:
        here 1.5 s of start-up against ~0.1 s of work).  Same per-frame products as calibrate() (bad-pixel repair, optional
        L.A.Cosmic, the same keywords).  timings: a dict that receives the wall seconds of the phases (read = FITS read + decode
        into the slab, compute, write = encode + FITS write)."""
        from .. import ops
        if len(raw_images) != len(cal_images):
            raise ValueError('raw_images and cal_images differ in length')
        t0 = time.perf_counter()
        try:
            slab, hdrs, ratios, peds = self.load_slab(raw_images)
        except TypeError:                                   # float64 and other raw types mixed: one frame at a time
            for src, dst in zip(raw_images, cal_images):
                self.calibrate(src, dst, delta_pix, None, fixcosmic)
            return
        import torch
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        cal = self.calibrate_slab(slab, ratios, peds)
        t_compute, t_write = 0.0, 0.0
        # the output files are written by a few threads while the next frame is repaired and encoded (fitsio.WritePool)
        self._write_pool = fitsio.shared_write_pool() if len(raw_images) > 1 else None
        try:
            t_compute, t_write = self._calibrate_files_loop(raw_images, cal_images, cal, hdrs, delta_pix, fixcosmic)
        finally:
            pool, self._write_pool = self._write_pool, None
            if pool is not None:
                tw = time.perf_counter()
                pool.wait()                                 # every file in place (or the first error) before this call returns
                t_write += time.perf_counter() - tw
        if timings is not None:
            timings.update(read=t1 - t0, compute=t_compute, write=t_write, total=time.perf_counter() - t0, frames=len(raw_images))

    def _calibrate_files_loop(self, raw_images, cal_images, cal, hdrs, delta_pix, fixcosmic):
        from .. import ops
        import torch
        t_compute, t_write = 0.0, 0.0
        for i, (src, dst) in enumerate(zip(raw_images, cal_images)):
            tc = time.perf_counter()
            odict = self._base_keywords()
            img = cal[i]
            if self._bpix is not None:
                img, st = ops.fix_badpix(img, self._mskdata, int(delta_pix), self._bpix._min_valid)
                nbad, nfixed, nnotfix = (int(x) for x in st.cpu().numpy())
                odict.update({'BPIXFILE': (self._master_bpix.name, 'Name of master bad pixel file used'),
                              'BPIXNBAD': (nbad, 'Total number of bad pixels in bad pixel file'),
                              'BPIX_MIN': (self._bpix._min_valid, 'Minimum number of good neighors needed'),
                              'BPIXDPIX': (int(delta_pix), 'Half height/width of collection region (pixels)'),
                              'BPIXNREM': (nnotfix, 'Number of bad pixels not corrected'),
                              'BPIXCORR': (nfixed > 0, 'True if any bad pixels were corrected'),
                              'BPIXNFIX': (nfixed, 'Number of bad pixels corrected')})
            if fixcosmic:                                       # ApCalibrate.py:490-497
                gain = self._get_gain(hdrs[i])
                if self._crfix is None:
                    from .ApFixCosmicRays import ApFixCosmicRays
                    self._crfix = ApFixCosmicRays(self._loglevel)
                img, crmask, _ = self._crfix.process_tensor(img, gain)
                odict['CR_CLEAN'] = (True, 'Has cosmic ray removal been performed?')
                odict['CR_NPIX'] = (int(crmask.sum()), 'Number of pixels modified by lacosmic.')
            torch.cuda.synchronize()
            tw = time.perf_counter()
            self._write_corrected_image(src, dst, img, odict)
            t_compute += tw - tc
            t_write += time.perf_counter() - tw
        return t_compute, t_write
