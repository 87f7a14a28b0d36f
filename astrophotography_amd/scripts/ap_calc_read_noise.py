#!/usr/bin/env python3
"""ap_calc_read_noise - detector read noise (e/pixel) from two raw bias frames and the gain
(reference: scripts/ap_calc_read_noise.py:41-84, 690-706)."""
import argparse
import logging


def command_line_opts(argv):
    parser = argparse.ArgumentParser(prog='ap_calc_read_noise',
                                     description='Estimate the detector read noise (e/pixel) from two raw bias images '
                                                 'and the electronic gain (e/ADU).')
    parser.add_argument('biasfile1', metavar='RAW_BIAS_1.FITS', help='First raw bias file.')
    parser.add_argument('biasfile2', metavar='RAW_BIAS_2.FITS', help='Second raw bias file.')
    p_gain = 'EGAIN'
    parser.add_argument('--gain', default=p_gain,
                        help=f'FITS keyword holding the gain (e/ADU) or a numerical gain value. Default: {p_gain}')
    parser.add_argument('--noclip', dest='sigmaclip', action='store_false', default=True,
                        help='Do NOT remove outlier pixels of either bias frame by sigma clipping.')
    parser.add_argument('--histplot', default=None, help='Histogram plot file (not available in this build).')
    parser.add_argument('-l', '--loglevel', default='INFO', help='Logging message level. Default: INFO')
    return parser.parse_args(argv)


def main(args=None):
    p_args = command_line_opts(args)
    import astrophotography_amd as ap
    calc = ap.ApCalcReadNoise(p_args.biasfile1, p_args.biasfile2, p_args.gain, p_args.loglevel)
    rn1 = calc.estimate_rn(p_args.sigmaclip, p_args.histplot)
    print(f'Estimated read noise is {rn1:.2f} electrons/pixel.')
    return 0


if __name__ == '__main__':
    try:
        status = main()
    except Exception:
        logging.getLogger(__name__).critical('Shutting down due to fatal error')
        raise
    else:
        raise SystemExit(status)
