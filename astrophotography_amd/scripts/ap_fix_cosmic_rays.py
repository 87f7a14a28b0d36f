#!/usr/bin/env python3
"""ap_fix_cosmic_rays - L.A.Cosmic cleaning of one FITS image (reference: scripts/ap_fix_cosmic_rays.py:36-100)."""
import argparse
import logging


def command_line_opts(argv):
    parser = argparse.ArgumentParser(prog='ap_fix_cosmic_rays',
                                     description='Clean cosmic rays from a CCD image using the L.A. Cosmic algorithm. The input image '
                                                 'should have already undergone bad pixel correction (e.g. using ap_fix_badpix.py) before '
                                                 'attempting cosmic ray correction.')
    parser.add_argument('input', metavar='INPUT.FITS', help='Path/name of the input FITS image.')
    parser.add_argument('output', metavar='OUTPUT.FITS', help='Path/name of the output cosmic-ray cleaned FITS image.')
    parser.add_argument('--crdiffim', metavar='CR_DIFFERENCE_IMG.FITS', default=None,
                        help='Name of the optional difference image: the CR cleaned image subtracted from the original image.')
    parser.add_argument('--crmaskim', metavar='CR_MASK_IMG.FITS', default=None,
                        help='Name of the optional CR mask image, non-zero at the location of the identified cosmic rays.')
    parser.add_argument('-l', '--loglevel', default='INFO', help='Logging message level. Default: INFO')
    return parser.parse_args(argv)


def main(args=None):
    p = command_line_opts(args)
    import astrophotography_amd as ap
    crfixer = ap.ApFixCosmicRays(p.loglevel)
    crfixer.process_file(p.input, p.output)
    if p.crdiffim is not None:
        crfixer.write_crdiff_img(p.crdiffim)
    if p.crmaskim is not None:
        crfixer.write_crmask_img(p.crmaskim)
    return 0


if __name__ == '__main__':
    try:
        status = main()
    except Exception:
        logging.getLogger(__name__).critical('Shutting down due to fatal error')
        raise
    else:
        raise SystemExit(status)
