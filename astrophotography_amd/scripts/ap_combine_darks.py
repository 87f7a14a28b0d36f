#!/usr/bin/env python3
"""ap_combine_darks - master dark/bias/flat from a directory of frames (reference: scripts/ap_combine_darks.py:48-474)."""
import argparse
import logging


def command_line_opts(argv):
    parser = argparse.ArgumentParser(prog='ap_combine_darks', description='Generates a master dark, bias or flat from '
                                                                          'all calibration FITS files in a directory.')
    parser.add_argument('rawcaldir', metavar='RAW_CAL_DIR', help='Directory holding the raw calibration frames.')
    parser.add_argument('master_filename', metavar='MASTER_CAL_FILENAME', help='Output master calibration file.')
    p_temptol, p_telescop, p_exclude = 0.5, 'UNKNOWN', 'master*'
    parser.add_argument('-l', '--loglevel', default='INFO', help='Logging message level. Default: INFO')
    parser.add_argument('--exclude', dest='exclude_pattern', default=p_exclude, metavar='FILE_PATTERN',
                        help=f'Unix-style pattern of files to skip. Default: "{p_exclude}"')
    parser.add_argument('--telescop', default=p_telescop, metavar='TELESCOPE_NAME',
                        help=f'TELESCOP value to write if the inputs have none. Default: {p_telescop}')
    parser.add_argument('--temptol', default=p_temptol, type=float, metavar='DEGREES_C',
                        help=f'Allowed |CCD-TEMP - SET-TEMP|. Default: {p_temptol} C.')
    return parser.parse_args(argv)


def main(args=None):
    p_args = command_line_opts(args)
    logger = logging.getLogger(__name__)
    import astrophotography_amd as ap
    try:
        mkcal = ap.ApMasterCal(p_args.rawcaldir, p_args.exclude_pattern, p_args.telescop, p_args.temptol, p_args.loglevel)
        mkcal.make_master(p_args.master_filename)
    except RuntimeError as rte:
        logger.error(f'Shutting down due to exception raised by ApMasterCal: {rte}')
        return 1
    return 0


if __name__ == '__main__':
    try:
        status = main()
    except Exception:
        logging.getLogger(__name__).critical('Shutting down due to fatal error')
        raise
    else:
        raise SystemExit(status)
