#!/usr/bin/env python3
"""ap_imarith - image (op) image|scalar arithmetic (reference: scripts/ap_imarith.py:40-115)."""
import argparse
import logging


def command_line_opts(argv):
    parser = argparse.ArgumentParser(prog='ap_imarith', description='ADD/SUB/MUL/DIV a FITS image with another image '
                                                                    'of the same size or with a scalar.')
    parser.add_argument('input_image', metavar='INPUT_IMAGE.FITS', help='First operand.')
    parser.add_argument('operation', metavar='OPERATION', help='One of ADD, SUB, MUL, DIV.')
    parser.add_argument('value', metavar='VALUE_OR_IMAGE', help='Scalar or second FITS image.')
    parser.add_argument('output_image', metavar='OUTPUT_IMAGE.FITS', help='Output file (overwritten).')
    parser.add_argument('--units', default=None, metavar='BUNIT', help='New BUNIT value for the output.')
    parser.add_argument('-l', '--loglevel', default='INFO', help='Logging message level. Default: INFO')
    return parser.parse_args(argv)


def main(args=None):
    p_args = command_line_opts(args)
    import astrophotography_amd as ap
    imarith = ap.ApImArith(p_args.loglevel)
    imarith.process_files(p_args.input_image, p_args.operation, p_args.value, p_args.output_image, p_args.units)
    return 0


if __name__ == '__main__':
    try:
        status = main()
    except Exception:
        logging.getLogger(__name__).critical('Shutting down due to fatal error')
        raise
    else:
        raise SystemExit(status)
