"""ApImageDifference / ApCalcReadNoise - read noise from two bias frames (reference:
scripts/ap_calc_read_noise.py:86-370 and :371-688).

    good   = both images inside their own 3-sigma clipped range  (two global clips, A3; :247-286)
    diff   = float64(image1) - float64(image2)                   (:122)
    sigma  = np.std(diff[good]);  read noise = gain * sigma / sqrt(2)   (:330-335, :553)

On the device: two `apgpu_sigclip_global` runs, two threshold masks, `apgpu_image_difference_f64` and one
more global-statistics pass over the difference (numpy's float64 summation order), nothing on the CPU.
"""
import math

import numpy as np

from . import _common


class ApImageDifference:
    """Difference of two images and its statistics over the pixels that are good in both."""

    def __init__(self, imdata1, imdata2, sigmaclip, loglevel, mask1=None, mask2=None):
        self._loglevel = loglevel
        self._logger = _common.make_logger('ApImageDifference', loglevel)
        imdata1, imdata2 = np.asarray(imdata1), np.asarray(imdata2)
        self._check_input_images(imdata1, imdata2)
        self._compute(imdata1, imdata2, bool(sigmaclip), mask1, mask2)

    def _check_input_images(self, imdata1, imdata2):
        if imdata1.shape != imdata2.shape:
            err_msg = ('Error, data array shapes do not match:'
                       f' First file={imdata1.shape}, second file={imdata2.shape}')
            self._logger.error(err_msg)
            raise RuntimeError(err_msg)
        if imdata1.dtype != imdata2.dtype:
            err_msg = f'Error, data types do not match: First file={imdata1.dtype}, second file={imdata2.dtype}'
            self._logger.error(err_msg)
            raise RuntimeError(err_msg)

    @staticmethod
    def _to_device(a):
        import torch
        from .. import ops
        if a.dtype == np.uint16:
            return ops.to_device_u16(a)
        if a.dtype == np.float32:
            return torch.from_numpy(np.ascontiguousarray(a)).cuda()
        raise TypeError(f'ApImageDifference supports uint16 and float32 images on the GPU, not {a.dtype}')

    def _clip_bad_mask(self, dev_img, host_dtype):
        """uint8 device mask of pixels OUTSIDE the image's own 3-sigma clipped range (:263-277)."""
        import torch
        from .. import ops
        sigma = 3.0
        s = ops.sigclip_global(dev_img, sigma=sigma, maxiters=5).cpu().numpy()
        med, std = float(s[1]), float(s[2])
        lo, hi = med - (sigma * std), med + (sigma * std)
        f32 = dev_img if dev_img.dtype == torch.float32 else (dev_img.view(torch.int16).to(torch.int32) & 0xFFFF).to(torch.float32)
        bad, _ = ops.threshold_mask(f32, lo, hi)          # good = (x >= lo) & (x <= hi)  <=>  not (x < lo or x > hi)
        return bad, lo, hi

    def _compute(self, im1, im2, sigmaclip, mask1, mask2):
        import torch
        from .. import ops
        d1, d2 = self._to_device(im1), self._to_device(im2)
        self._d1, self._d2 = d1, d2
        bad1 = bad2 = None
        if sigmaclip:
            self._logger.debug('Generating a good pixel mask using sigma=3.0 clipping on input image data values.')
            bad1, lo1, hi1 = self._clip_bad_mask(d1, im1.dtype)
            bad2, lo2, hi2 = self._clip_bad_mask(d2, im2.dtype)
            self._logger.debug(f'Good pixels in the first image have pixel values between {lo1:.2f} and {hi1:.2f} ADU.')
            self._logger.debug(f'Good pixels in the second image have pixel values between {lo2:.2f} and {hi2:.2f} ADU.')
        elif mask1 is not None or mask2 is not None:
            if mask1 is not None and mask2 is not None and np.asarray(mask1).shape != np.asarray(mask2).shape:
                err_msg = f'Error, mask shapes do not match. mask1 is {np.asarray(mask1).shape}, while mask2 is {np.asarray(mask2).shape}'
                self._logger.error(err_msg)
                raise RuntimeError(err_msg)
            bad1 = None if mask1 is None else torch.from_numpy(np.ascontiguousarray(np.asarray(mask1) != 0).view(np.uint8)).cuda()
            bad2 = None if mask2 is None else torch.from_numpy(np.ascontiguousarray(np.asarray(mask2) != 0).view(np.uint8)).cuda()
        self._diff_dev = ops.image_difference(d1, d2, bad1, bad2)       # NaN where a pixel is bad in either image
        st = ops.sigclip_global(self._diff_dev, sigma=1e300, maxiters=1).cpu().numpy()
        self._stats = dict(mean=st[0], median=st[1], stddev=st[2], min=st[7], max=st[8])
        self._numpix = im1.size
        self._numgood = int(st[6])
        self._logger.debug(f'Final good pixel mask has {self._numgood} good pixels out of {self._numpix} pixels '
                           f'({self._numpix - self._numgood} bad).')
        self._bad_dev = (bad1, bad2)

    def data(self):
        """The float64 difference image of ALL pixels (as the reference returns it)."""
        from .. import ops
        return ops.image_difference(self._d1, self._d2).cpu().numpy()

    def good_pixel_mask(self):
        import torch
        return torch.isfinite(self._diff_dev).cpu().numpy()       # True where the pixel is good in both images

    def stddev(self):
        return self._stats['stddev']

    def min(self):
        return self._stats['min']

    def max(self):
        return self._stats['max']

    def mean(self):
        return self._stats['mean']

    def median(self):
        return self._stats['median']

    def numpix(self):
        return self._numgood, self._numpix


class ApCalcReadNoise:
    def __init__(self, biasfile1, biasfile2, gain, loglevel):
        self._loglevel = loglevel
        self._biasfile1 = biasfile1
        self._biasfile2 = biasfile2
        self._gaininfo = gain
        self._logger = _common.make_logger('ApCalcReadNoise', loglevel)
        self._logger.debug(f'Initialized an ApCalcReadNoise instance with biasfile1={biasfile1}, biasfile1={biasfile2}, '
                           f'gain={gain}, and loglevel={loglevel}')

    @staticmethod
    def _isfloat(value_str):
        try:
            float(value_str)
            return True
        except (TypeError, ValueError):
            return False

    def _select_gain(self, hdr1, hdr2):
        """A number given at construction wins; otherwise the keyword must exist in both headers and agree
        to 0.001 e/ADU (ap_calc_read_noise.py:634-688)."""
        if self._isfloat(self._gaininfo):
            return float(self._gaininfo)
        gain1 = float(hdr1[self._gaininfo]) if self._gaininfo in hdr1 else None
        gain2 = float(hdr2[self._gaininfo]) if self._gaininfo in hdr2 else None
        if gain1 is None or gain2 is None:
            err_msg = f'Error, {self._gaininfo} gain keyword not found in'
            if gain1 is None and gain2 is None:
                err_msg += ' both FITS files.'
            elif gain1 is None:
                err_msg += ' the first FITS file.'
            else:
                err_msg += ' the second FITS file.'
            self._logger.error(err_msg)
            raise RuntimeError(err_msg)
        tolerance = 0.001
        if math.fabs(gain1 - gain2) > tolerance:
            err_msg = f'Error, gains differ by more than {tolerance:.3f} e/ADU, where gain1={gain1:.3f}, gain2={gain2:.3f}.'
            self._logger.error(err_msg)
            raise RuntimeError(err_msg)
        return gain1

    def estimate_rn(self, sigmaclip, histplot=None):
        if histplot is not None:
            raise RuntimeError('Histogram plotting (matplotlib) is outside the scope of the MI355X path; pass histplot=None.')
        data1, hdr1, _ = _common.read_fits(self._logger, self._biasfile1)
        data2, hdr2, _ = _common.read_fits(self._logger, self._biasfile2)
        if data1.shape != data2.shape:
            err_msg = f'Error, data array shapes do not match: First file={data1.shape}, second file={data2.shape}'
            self._logger.error(err_msg)
            raise RuntimeError(err_msg)
        self._gain = self._select_gain(hdr1, hdr2)
        self._logger.info(f'Adopted gain is {self._gain:.2f} electrons/ADU.')
        im_diff = ApImageDifference(data1, data2, sigmaclip, self._loglevel, mask1=None, mask2=None)
        stddev = im_diff.stddev()
        npix_good, npix_total = im_diff.numpix()
        pct_bad = 100 * (npix_total - npix_good) / npix_total
        self._logger.info(f'Standard deviation={stddev:.2f} ADU using {npix_good}/{npix_total} pixels ({pct_bad:.3f} % bad).')
        read_noise = self._gain * stddev / math.sqrt(2)
        self._logger.info(f'Estimated read noise is {read_noise:.2f} e/pixel')
        return read_noise
