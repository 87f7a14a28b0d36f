#!/usr/bin/env python3
"""ap_calibrate - bias/dark/flat (and bad pixel) calibration of one raw frame on the GPU.

Same positional arguments and flags as the reference script (scripts/ap_calibrate.py:40-122);
``--fixcosmic`` runs L.A.Cosmic (ApFixCosmicRays, csrc/lacosmic.hip) after the bad-pixel repair, as in the reference.

Not in the reference: ``--batch LIST`` calibrates every "RAW.fits CALIBRATED.fits" pair of the text file LIST in THIS process
(ApCalibrate.calibrate_files: masters read once, one slab, one launch; ``-`` in place of the two image arguments).  The
reference's driver starts one interpreter per frame (scripts/calibrate_all.sh:406-411), which here costs ~1.5 s of start-up per
frame against ~0.1 s of work (profiles/r06/frame_path.txt).
"""
import argparse
import logging
import sys


def command_line_opts(argv):
    parser = argparse.ArgumentParser(prog='ap_calibrate',
                                     description='Bias and dark subtraction, optional flat fielding and bad pixel '
                                                 'correction of a raw CCD frame (MI355X HIP kernels).')
    parser.add_argument('raw_image', metavar='INPUT_IMAGE.FITS', help='Raw input FITS image.')
    parser.add_argument('master_bias', metavar='MBIAS.FITS', help='Master bias frame.')
    parser.add_argument('master_dark', metavar='MDARK.FITS', help='Master dark frame.')
    parser.add_argument('calibrated_image', metavar='CALIBRATED_IMAGE.FITS', help='Output file (overwritten).')
    p_delta = 2
    parser.add_argument('--master_flat', metavar='MFLAT.FITS', default=None, help='Master flat for this filter.')
    parser.add_argument('--master_badpix', metavar='BADPIX.FITS', default=None, help='Master bad pixel mask.')
    parser.add_argument('--normflat', metavar='NORMALIZED_FLAT.FITS', default=None,
                        help='Also write the normalised flat field to this file.')
    parser.add_argument('--deltapix', default=p_delta, type=int, metavar='NPIX',
                        help=f'Half-size of the box good neighbours are drawn from. Default: {p_delta}')
    parser.add_argument('--fixcosmic', default=False, action='store_true', help='If specified, cosmic rays will be removed with the L.A.Cosmic algorithm after bad pixel correction.')
    parser.add_argument('--dark_still_biased', default=False, action='store_true',
                        help='The master dark has NOT had the bias subtracted yet.')
    parser.add_argument('--batch', metavar='LIST.TXT', default=None,
                        help='Calibrate every "RAW.fits CALIBRATED.fits" pair listed in this text file in one process '
                             '(INPUT_IMAGE.FITS and CALIBRATED_IMAGE.FITS are then ignored: pass -). Chunks of --batch_size frames.')
    parser.add_argument('--batch_size', default=16, type=int, metavar='N', help='Frames per slab in --batch mode. Default: 16')
    parser.add_argument('-l', '--loglevel', default='INFO', help='Logging message level. Default: INFO')
    return parser.parse_args(argv)


def main(args=None):
    p_args = command_line_opts(args)
    import astrophotography_amd as ap
    calibrator = ap.ApCalibrate(p_args.master_bias, p_args.master_dark, p_args.master_flat, p_args.master_badpix,
                                p_args.loglevel, p_args.dark_still_biased)
    if p_args.batch is not None:
        with open(p_args.batch) as fh:
            pairs = [ln.split() for ln in fh if ln.strip() and not ln.lstrip().startswith('#')]
        if any(len(pr) != 2 for pr in pairs):
            raise RuntimeError(f'{p_args.batch}: every line must hold "RAW.fits CALIBRATED.fits"')
        n = max(1, int(p_args.batch_size))
        for i in range(0, len(pairs), n):
            chunk = pairs[i:i + n]
            calibrator.calibrate_files([c[0] for c in chunk], [c[1] for c in chunk], p_args.deltapix, p_args.fixcosmic)
        return 0
    calibrator.calibrate(p_args.raw_image, p_args.calibrated_image, p_args.deltapix, p_args.normflat, p_args.fixcosmic)
    return 0


if __name__ == '__main__':
    try:
        status = main()
    except Exception:
        logging.getLogger(__name__).critical('Shutting down due to fatal error')
        raise
    else:
        raise SystemExit(status)
