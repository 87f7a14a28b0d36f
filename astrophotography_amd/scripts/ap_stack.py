#!/usr/bin/env python3
"""ap_stack - sigma-clipped / median / mean stack of N FITS frames, optionally calibrating them on the fly
(new front-end for ApStack; the reference stacks light frames with the external SWarp program)."""
import argparse
import logging


def command_line_opts(argv):
    parser = argparse.ArgumentParser(prog='ap_stack', description='Stack N registered frames into one image on the GPU.')
    parser.add_argument('output_image', metavar='OUTPUT_IMAGE.FITS', help='Output stacked image (overwritten).')
    parser.add_argument('input_images', metavar='INPUT_IMAGE.FITS', nargs='+', help='Frames to stack.')
    parser.add_argument('--method', default='sigclip', choices=['sigclip', 'median', 'mean'])
    parser.add_argument('--sigma', default=3.0, type=float, metavar='NSIGMA')
    parser.add_argument('--maxiters', default=5, type=int, metavar='N', help='-1 = until convergence')
    parser.add_argument('--cenfunc', default='median', choices=['median', 'mean'])
    parser.add_argument('--stdfunc', default='std', choices=['std', 'mad_std'])
    parser.add_argument('--master_bias', default=None, metavar='MBIAS.FITS', help='Calibrate raw inputs on the fly ...')
    parser.add_argument('--master_dark', default=None, metavar='MDARK.FITS')
    parser.add_argument('--master_flat', default=None, metavar='MFLAT.FITS')
    parser.add_argument('--dark_still_biased', default=False, action='store_true')
    parser.add_argument('-l', '--loglevel', default='INFO', help='Logging message level. Default: INFO')
    return parser.parse_args(argv)


def main(args=None):
    p = command_line_opts(args)
    import astrophotography_amd as ap
    calibrator = None
    if p.master_bias or p.master_dark:
        if not (p.master_bias and p.master_dark):
            raise RuntimeError('Fused calibration needs both --master_bias and --master_dark.')
        calibrator = ap.ApCalibrate(p.master_bias, p.master_dark, p.master_flat, None, p.loglevel, p.dark_still_biased)
    stacker = ap.ApStack(p.loglevel, sigma=p.sigma, maxiters=None if p.maxiters < 0 else p.maxiters, cenfunc=p.cenfunc,
                         stdfunc=p.stdfunc)
    stacker.stack_files(p.input_images, p.output_image, method=p.method, calibrator=calibrator)
    return 0


if __name__ == '__main__':
    try:
        status = main()
    except Exception:
        logging.getLogger(__name__).critical('Shutting down due to fatal error')
        raise
    else:
        raise SystemExit(status)
