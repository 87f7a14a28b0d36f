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

    _GAIN_TOLERANCE = 0.001          # e/ADU: the two headers must agree this well (ap_calc_read_noise.py:634-688)

    def _fail(self, err_msg):
        self._logger.error(err_msg)
        raise RuntimeError(err_msg)

    def _gain_from(self, headers):
        """The gain in e/ADU: a number passed at construction is taken as it is; anything else names a header keyword,
        which both bias frames must carry with the same value (to _GAIN_TOLERANCE)."""
        spec = self._gaininfo
        try:
            return float(spec)
        except (TypeError, ValueError):
            pass
        found = [float(h[spec]) if spec in h else None for h in headers]
        absent = tuple(g is None for g in found)
        if any(absent):
            where = {(True, True): 'both FITS files.', (True, False): 'the first FITS file.',
                     (False, True): 'the second FITS file.'}[absent]
            self._fail(f'Error, {spec} gain keyword not found in {where}')
        spread = abs(found[0] - found[1])
        if spread > self._GAIN_TOLERANCE:
            self._fail(f'Error, gains differ by more than {self._GAIN_TOLERANCE:.3f} e/ADU, '
                       f'where gain1={found[0]:.3f}, gain2={found[1]:.3f}.')
        return found[0]

    def estimate_rn(self, sigmaclip, histplot=None):
        """Read noise in e/pixel = gain * std(bias1 - bias2 over the good pixels) / sqrt(2) (ap_calc_read_noise.py:507-556)."""
        if histplot is not None:
            raise RuntimeError('Histogram plotting (matplotlib) is outside the scope of the MI355X path; pass histplot=None.')
        frames = [_common.read_fits(self._logger, f)[:2] for f in (self._biasfile1, self._biasfile2)]
        (data1, hdr1), (data2, hdr2) = frames
        if data1.shape != data2.shape:
            self._fail(f'Error, data array shapes do not match: First file={data1.shape}, second file={data2.shape}')
        self._gain = self._gain_from((hdr1, hdr2))
        self._logger.info(f'Adopted gain is {self._gain:.2f} electrons/ADU.')
        # everything numeric happens on the device inside ApImageDifference
        diff = ApImageDifference(data1, data2, sigmaclip, self._loglevel, mask1=None, mask2=None)
        sigma_adu = diff.stddev()
        good, total = diff.numpix()
        self._logger.info(f'Standard deviation={sigma_adu:.2f} ADU using {good}/{total} pixels '
                          f'({100 * (total - good) / total:.3f} % bad).')
        read_noise = self._gain * sigma_adu / math.sqrt(2)
        self._logger.info(f'Estimated read noise is {read_noise:.2f} e/pixel')
        return read_noise
