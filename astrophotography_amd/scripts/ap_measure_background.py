#!/usr/bin/env python3
"""ap_measure_background - large-scale sky background of a calibrated image (reference: scripts/ap_measure_background.py:40-180).
Same positional arguments and flags; the result is subtracted with ap_imarith.py SUB (calibrate_all.sh:440-460)."""
import argparse
import logging


def command_line_opts(argv):
    parser = argparse.ArgumentParser(prog='ap_measure_background',
                                     description='Measures and outputs large scale non-uniform sky backgrounds left by imperfect '
                                                 'bias/dark/flat calibration. Successful background estimation may require you to experiment '
                                                 'with the options that control the background region box size and filtering parameters.')
    parser.add_argument('input', metavar='INPUT_IMAGE.FITS', help='Path/name of the input image that the background will be measured in.')
    parser.add_argument('outputbg', metavar='OUTPUT_BG.FITS',
                        help='Path/name of the output estimate of the large scale background. This can be subtracted from the input '
                             'image using ap_imarith.py.')
    parser.add_argument('--srclist', metavar='SRCLIST.FITS', default=None,
                        help='Optional source list to use to exclude stars when generating the background estimate (not yet used, as in '
                             'the reference).')
    parser.add_argument('--nbg_cols', metavar='NUM_BGCOLS', type=int, default=16,
                        help='Number of regions to split the image into width-wise to assess the local background in. Default: 16')
    parser.add_argument('--nbg_rows', metavar='NUM_BGROWS', type=int, default=16,
                        help='Number of regions to split the image into height-wise to assess the local background in. Default: 16')
    parser.add_argument('--min_bgwidth', metavar='MIN_BGWIDTH', type=int, default=48,
                        help='Minimum width in pixels of a background region. Default: 48 pixels')
    parser.add_argument('--min_bgheight', metavar='MIN_BGHEIGHT', type=int, default=48,
                        help='Minimum height in pixels of a background region. Default: 48 pixels')
    parser.add_argument('--bg_filter_width', metavar='FILTER_WIDTH', type=int, default=3,
                        help='Background2D filter size parameter, used to median filter the course background values. Default: 3 course boxes.')
    parser.add_argument('--bg_badbox_pctile', metavar='PERCENTILE', type=float, default=25.0,
                        help='Background2D exclude_percentile parameter: course background cells with more than this percentage of '
                             'their pixels excluded or masked are marked bad. Default: 25.00 percent.')
    parser.add_argument('--bg_sigmaclip', metavar='NSIGMA', type=float, default=3.0,
                        help='Background2D sigma_clip parameter: sigma clipping of the pixel values within each course background box. '
                             'Default: 3.00 sigma.')
    parser.add_argument('-l', '--loglevel', default='INFO', help='Logging message level. Default: INFO')
    return parser.parse_args(argv)


def main(args=None):
    p = command_line_opts(args)
    import astrophotography_amd as ap
    meas_bg = ap.ApMeasureBackground(p.loglevel)
    meas_bg.process_files(p.input, p.srclist, p.nbg_rows, p.nbg_cols, p.min_bgheight, p.min_bgwidth, p.bg_filter_width,
                          p.bg_badbox_pctile, p.bg_sigmaclip)
    meas_bg.write_bgimage(p.outputbg)
    return 0


if __name__ == '__main__':
    try:
        status = main()
    except Exception:
        logging.getLogger(__name__).critical('Shutting down due to fatal error')
        raise
    else:
        raise SystemExit(status)
