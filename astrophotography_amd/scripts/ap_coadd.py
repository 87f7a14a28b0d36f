#!/usr/bin/env python3
"""ap_coadd - Lanczos-3 resample + co-add of registered frames on the GPU (the role SWarp plays in the
reference's scripts/resample_all.sh:330-342).  Transforms come from a YAML file:

    transforms:
      frame-0001.fits: [1.0, 0.0, 0.0, 0.0, 1.0, 0.0]     # xin = a0*x + a1*y + a2 ; yin = a3*x + a4*y + a5
      frame-0002.fits: [0.99999, -0.0035, 1.25, 0.0035, 0.99999, -0.75]
"""
import argparse
import logging
import os


def command_line_opts(argv):
    parser = argparse.ArgumentParser(prog='ap_coadd', description='Resample registered frames onto one grid and combine them.')
    parser.add_argument('output_image', metavar='OUTPUT_IMAGE.FITS', help='Output co-added image (overwritten).')
    parser.add_argument('input_images', metavar='INPUT_IMAGE.FITS', nargs='+', help='Calibrated frames to combine.')
    parser.add_argument('--transforms', default=None, metavar='TRANSFORMS.YML',
                        help='Per-file 2x3 affine transforms. Default: register through the TAN WCS of the file headers.')
    parser.add_argument('--center', default=None, metavar='RA,DEC', help='Output tangent point in degrees (WCS mode).')
    parser.add_argument('--pixelscale', default=None, type=float, metavar='ARCSEC', help='Output pixel scale (WCS mode).')
    parser.add_argument('--combine', default='MEDIAN', choices=['MEDIAN', 'AVERAGE', 'WEIGHTED', 'SUM', 'CLIPPED'],
                        help='Combine type (resample_all.sh add modes 0/1/2 = MEDIAN/WEIGHTED/SUM). Default: MEDIAN')
    parser.add_argument('--weight_image', default=None, metavar='WEIGHTS.FITS', help='Optional output weight image.')
    parser.add_argument('--badpix', default=None, metavar='BADPIX.FITS', help='Optional bad pixel mask shared by the inputs.')
    parser.add_argument('--image_size', default=None, metavar='NX,NY', help='Output size (default: input size).')
    parser.add_argument('--oversampling', default=1, type=int, metavar='N',
                        help='Sub-samples per output pixel and axis (SWarp OVERSAMPLING; resample_all.sh uses 4). Default: 1')
    parser.add_argument('--gain_keyword', default='EGAIN', metavar='KEYWORD',
                        help='Header keyword with the detector gain in e-/ADU (SWarp GAIN_KEYWORD). Default: EGAIN')
    parser.add_argument('--sigma', default=3.0, type=float, metavar='NSIGMA', help='CLIPPED only.')
    parser.add_argument('--maxiters', default=5, type=int, metavar='N', help='CLIPPED only; -1 = until convergence.')
    parser.add_argument('-l', '--loglevel', default='INFO', help='Logging message level. Default: INFO')
    return parser.parse_args(argv)


def main(args=None):
    p = command_line_opts(args)
    import yaml
    from astrophotography_amd.core.ApResample import ApResample
    affines = None
    table = None
    if p.transforms is not None:
        with open(p.transforms) as fh:
            doc = yaml.safe_load(fh) or {}
        table = doc.get('transforms') or {}
        affines = []
    for f in (p.input_images if table is not None else []):
        key = f if f in table else os.path.basename(f)
        if key not in table:
            raise RuntimeError(f'Error, no transform for {f} in {p.transforms}.')
        if len(table[key]) != 6:
            raise RuntimeError(f'Error, transform of {key} must have 6 coefficients.')
        affines.append([float(v) for v in table[key]])
    out_shape = None
    if p.image_size:
        nx, ny = (int(v) for v in p.image_size.split(','))
        out_shape = (ny, nx)
    rs = ApResample(p.loglevel, combine=p.combine, sigma=p.sigma, maxiters=None if p.maxiters < 0 else p.maxiters,
                    oversampling=p.oversampling, gain_keyword=p.gain_keyword)
    center = None
    if p.center:
        center = tuple(float(v) for v in p.center.split(','))
    rs.coadd_files(p.input_images, affines, p.output_image, weight_file=p.weight_image, mask_file=p.badpix, out_shape=out_shape,
                   center=center, pixscale=p.pixelscale)
    return 0


if __name__ == '__main__':
    try:
        status = main()
    except Exception:
        logging.getLogger(__name__).critical('Shutting down due to fatal error')
        raise
    else:
        raise SystemExit(status)
