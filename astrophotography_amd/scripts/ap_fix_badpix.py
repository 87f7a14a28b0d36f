#!/usr/bin/env python3
"""ap_fix_badpix - median-of-good-neighbours repair of masked pixels (reference: scripts/ap_fix_badpix.py:35-100)."""
import argparse
import logging


def command_line_opts(argv):
    parser = argparse.ArgumentParser(prog='ap_fix_badpix', description='Replace bad pixels by the median of the good '
                                                                       'pixels around them.')
    parser.add_argument('raw_image', metavar='INPUT_IMAGE.FITS', help='Image with bad pixels.')
    parser.add_argument('master_badpix', metavar='BADPIX.FITS', help='Bad pixel mask (non-zero = bad).')
    parser.add_argument('fixed_image', metavar='OUTPUT_IMAGE.FITS', help='Output file (overwritten).')
    p_delta = 2
    parser.add_argument('--deltapix', default=p_delta, type=int, metavar='NPIX',
                        help=f'Half-size of the box good neighbours are drawn from. Default: {p_delta}')
    parser.add_argument('-l', '--loglevel', default='INFO', help='Logging message level. Default: INFO')
    return parser.parse_args(argv)


def main(args=None):
    p_args = command_line_opts(args)
    import astrophotography_amd as ap
    fixer = ap.ApFixBadPixels(p_args.loglevel)
    fixer.fix_files(p_args.raw_image, p_args.master_badpix, p_args.fixed_image, p_args.deltapix)
    return 0


if __name__ == '__main__':
    try:
        status = main()
    except Exception:
        logging.getLogger(__name__).critical('Shutting down due to fatal error')
        raise
    else:
        raise SystemExit(status)
