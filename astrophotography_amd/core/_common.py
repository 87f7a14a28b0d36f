"""Helpers shared by the Ap* shells (the reference copy-pastes these into every class:
_initialize_logger / _check_file_exists / _read_fits / _remove_pedestal_kw, e.g.
core/ApCalibrate.py:115-126, 230-258, 260-346)."""
import logging
from pathlib import Path

import numpy as np

from .. import fitsio

LOG_FORMAT = '%(asctime)s | %(name)s | %(levelname)s | %(message)s'


def make_logger(name, loglevel):
    """Per-class logger with the reference's format; raises ValueError for a bad level
    (core/ApCalibrate.py:230-258).  Unlike the reference, constructing a class twice does not add a
    second handler (duplicate lines)."""
    numeric_level = getattr(logging, str(loglevel).upper(), None)
    if not isinstance(numeric_level, int):
        raise ValueError('Invalid log level: {}'.format(loglevel))
    logger = logging.getLogger(name)
    logger.setLevel(numeric_level)
    if not logger.handlers:
        ch = logging.StreamHandler()
        ch.setFormatter(logging.Formatter(LOG_FORMAT))
        logger.addHandler(ch)
    for h in logger.handlers:
        h.setLevel(numeric_level)
    logger.propagate = False
    return logger


def check_file_exists(logger, filename):
    p = Path(filename).expanduser()
    if not p.exists():
        err_msg = f'Cannot find {filename}. Not a valid path or file.'
        logger.error(err_msg)
        raise RuntimeError(err_msg)
    return p


def read_fits(logger, image_filename, to_float32=False):
    """Primary-HDU data + header as the reference's _read_fits does (uint handling, 3-D rejection,
    PEDESTAL *added* to the data; core/ApCalibrate.py:260-328).

    to_float32=True is ApCalibrate's variant (integers -> float32, ApCalibrate.py:304-307); the other
    classes keep the file dtype (core/ApFindBadPixels.py:262-323).  Returns (data, header, pedestal)
    where pedestal is the value that was added (0.0 if none)."""
    image_filename = check_file_exists(logger, image_filename)
    logger.info('Loading extension {} of FITS file {}'.format(0, image_filename))
    data, hdr = fitsio.read(str(image_filename))
    ndim = hdr['NAXIS']
    if ndim == 3:
        logger.error('Error, 3-D handling has not been implemented yet.')
        raise SystemExit(1)
    if data is None or data.ndim != 2:
        raise RuntimeError(f'{image_filename}: expected a 2-D primary image, found NAXIS={ndim}.')
    if to_float32 and not np.issubdtype(data.dtype, np.floating):
        orig = data.dtype
        data = data.astype(np.float32)
        logger.debug(f'  Converted data type from {orig} to float32')
    pedestal = 0.0
    if 'PEDESTAL' in hdr:
        pedestal = float(hdr['PEDESTAL'])
        if pedestal != 0:
            logger.debug(f'Removing a PEDESTAL value of {pedestal} ADU.')
            if np.issubdtype(data.dtype, np.floating):
                data = data + data.dtype.type(pedestal)
            else:
                # numpy refuses `uint16 += float` (same_kind cast) - the reference raises here too
                raise TypeError(f'Cannot add PEDESTAL={pedestal} to integer data of type {data.dtype}.')
    return data, hdr, pedestal


def remove_pedestal_kw(logger, hdr):
    if 'PEDESTAL' in hdr:
        logger.debug('Removing PEDESTAL keyword from FITS header.')
        del hdr['PEDESTAL']
