"""ApImArith - host shell over apgpu_imarith (reference: core/ApImArith.py).

``process_files(inp_img, operation, value, out_img, units)`` (:255-346): image (op) image or image (op)
scalar with ADD/SUB/MUL/DIV, result in the dtype of the first image, BUNIT and two HISTORY cards.
Error behaviour follows the reference: ValueError for an unknown operation or an operand that is
neither a float nor an existing FITS file of the same shape; uint16 images accept only image
operands and ADD/SUB/MUL (numpy raises UFuncTypeError - a TypeError - for the rest).
"""
from datetime import datetime
from pathlib import Path

import numpy as np

from .. import __version__, fitsio
from . import _common


class ApImArith:
    def __init__(self, loglevel):
        self._name = 'ApImArith'
        self._allowed_ops = ['ADD', 'SUB', 'MUL', 'DIV']        # ApImArith.py:34
        self._loglevel = loglevel
        self._logger = _common.make_logger(self._name, loglevel)
        self._logger.debug(f'{self._name} instance constructed.')

    @staticmethod
    def _is_float(value):
        try:
            float(value)
            return True
        except (TypeError, ValueError):
            return False

    def _sanitize_operation(self, operation):
        cleaned = operation.strip().upper()
        if cleaned not in self._allowed_ops:
            raise ValueError(f'Error, input operation {cleaned} is not one of the allowed operations: {self._allowed_ops}')
        return cleaned

    def apply(self, data1, operation, data2):
        """Array form: numpy array (op) numpy array | float -> numpy array of data1's dtype."""
        import torch
        from .. import ops
        from .._lib import ApGpuError
        operation = self._sanitize_operation(operation)
        data1 = np.ascontiguousarray(data1)
        if data1.dtype in (np.float32, np.float64):
            a = torch.from_numpy(data1).cuda()
        elif data1.dtype == np.uint16:
            a = ops.to_device_u16(data1)
        else:
            raise TypeError(f'ApImArith supports float32, float64 and uint16 images on the GPU, not {data1.dtype}')
        if isinstance(data2, np.ndarray):
            if data1.shape != data2.shape:
                raise RuntimeError('Error, the dimension of the second data array does not match the first.'
                                   f' First image shape: {data1.shape}, second image shape: {data2.shape}')
            if data2.dtype != data1.dtype:
                if data1.dtype.kind == 'f':
                    # numpy computes in the promoted type and stores in data1's dtype (out=zeros_like(data1)):
                    # a float64 operand is passed as it is, anything else widens exactly to data1's dtype
                    if data2.dtype != np.float64:
                        data2 = data2.astype(data1.dtype if data2.dtype.itemsize <= 2 or data1.dtype == np.float64 else np.float64)
                else:
                    raise TypeError(f'Cannot cast {data2.dtype} operand to {data1.dtype} (same_kind)')
            b = torch.from_numpy(np.ascontiguousarray(data2)).cuda() if data1.dtype.kind == 'f' else ops.to_device_u16(data2)
        else:
            b = float(data2)
        try:
            out = ops.imarith(a, operation, b)
        except ApGpuError as e:
            if e.code == -2:            # numpy: UFuncTypeError (a TypeError) for u16 (op) scalar and u16 DIV
                raise TypeError(str(e))
            raise
        if out.dtype == torch.uint16:
            return out.view(torch.int16).cpu().numpy().view(np.uint16)
        return out.cpu().numpy()

    def _write_corrected_image(self, inpdata_file, outdata_file, odata, ounits, ohistory_str):
        _common.check_file_exists(self._logger, inpdata_file)
        _, hdr = fitsio.read(str(inpdata_file), want_data=False)
        _common.remove_pedestal_kw(self._logger, hdr)
        if ounits is not None:
            hdr['BUNIT'] = (ounits, 'Pixel value units')
        tnow = datetime.now().isoformat(timespec='milliseconds')
        hdr['HISTORY'] = f'Applied {self._name} {__version__} at {tnow}'
        hdr['HISTORY'] = ohistory_str
        fitsio.write(str(outdata_file), odata, hdr, overwrite=True)
        self._logger.info(f'Wrote modified data to {outdata_file}')

    def process_files(self, inp_img, operation, value, out_img, units):
        operation = self._sanitize_operation(operation)
        data1, _, _ = _common.read_fits(self._logger, inp_img)
        if self._is_float(value):
            data2 = float(value)
            second_data = f'scalar {data2}'
            value_str = f'{data2}'
        else:
            if not Path(value).exists():
                raise ValueError(f'Error, {value} is not a scalar or a valid file path.')
            try:
                data2, _, _ = _common.read_fits(self._logger, value)
                second_data = 'array'
                if data1.shape != data2.shape:
                    raise RuntimeError('shape mismatch')
            except Exception:
                raise ValueError(f'Error, {value} is not a valid FITS file.')
            value_str = Path(value).name
        result = self.apply(data1, operation, data2)
        verb = {'ADD': 'Added {} to input image', 'SUB': 'Subtracted {} from input image',
                'MUL': 'Multiplied input image by {}', 'DIV': 'Divided input image by {}'}[operation]
        self._logger.info(verb.format(second_data))
        ohistory_str = f'{self._name} input1 operation input2 are: {Path(inp_img).name} {operation} {value_str}'
        self._write_corrected_image(inp_img, out_img, result, units, ohistory_str)
        self._logger.debug('File processing completed.')
