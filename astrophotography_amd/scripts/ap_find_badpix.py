#!/usr/bin/env python3
"""ap_find_badpix - bad pixel mask from a master dark (reference: scripts/ap_find_badpix.py:35-100)."""
import argparse
import logging


def command_line_opts(argv):
    parser = argparse.ArgumentParser(prog='ap_find_badpix',
                                     description='Find bad pixels in a master dark/bias by global sigma clipping; '
                                                 'optionally add user-defined bad columns/rows/rectangles.')
    parser.add_argument('masterdark', metavar='IN_MASTER_DARK.FITS', help='Input master dark or bias.')
    parser.add_argument('badpixfile', metavar='OUT_BADPIX.FITS', help='Output bad pixel mask (overwritten).')
    p_sigma = 4.0
    parser.add_argument('--sigma', metavar='NSIGMA', default=p_sigma, type=float,
                        help=f'Number of clipped standard deviations from the median. Default: {p_sigma}')
    parser.add_argument('--user_badpix', metavar='USER_BADPIX.YML', default=None,
                        help='YAML file with bad_columns / bad_rows / bad_rectangles (1-based, inclusive).')
    parser.add_argument('-l', '--loglevel', default='INFO', help='Logging message level. Default: INFO')
    return parser.parse_args(argv)


def main(args=None):
    p_args = command_line_opts(args)
    import astrophotography_amd as ap
    bpix = ap.ApFindBadPixels(p_args.masterdark, p_args.sigma, p_args.loglevel)
    if p_args.user_badpix is not None:
        bpix.add_user_badpix(p_args.user_badpix)
    bpix.write_mask(p_args.badpixfile)
    return 0


if __name__ == '__main__':
    try:
        status = main()
    except Exception:
        logging.getLogger(__name__).critical('Shutting down due to fatal error')
        raise
    else:
        raise SystemExit(status)
