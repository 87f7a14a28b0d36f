"""Minimal FITS celestial WCS (gnomonic / TAN projection) for the registration step.

The reference registers its frames with astrometry.net and hands the navigated files to SWarp with
``PROJECTION_TYPE TAN`` (scripts/resample_all.sh:123-131, 330-342): every file carries CTYPE = RA---TAN /
DEC--TAN, CRPIX, CRVAL and a CD (or CDELT + PC / CROTA2) matrix.  This module turns such headers into
pixel <-> sky maps (FITS WCS paper II, zenithal gnomonic projection; pinned against astropy.wcs by golden
vectors G10) and into the per-tile affine transforms the resample kernel consumes.  Host-side float64 numpy.
SIP / TPV distortion terms are not supported (a header that carries them is refused).
"""
import numpy as np

D2R = np.pi / 180.0


class TanWcs:
    """0-based pixel (x = column, y = row) <-> (ra, dec) in degrees for a RA---TAN / DEC--TAN header."""

    def __init__(self, crpix, crval, cd):
        self.crpix = np.asarray(crpix, np.float64)          # FITS 1-based reference pixel (CRPIX1, CRPIX2)
        self.crval = np.asarray(crval, np.float64)          # (ra0, dec0) degrees
        self.cd = np.asarray(cd, np.float64).reshape(2, 2)  # degrees per pixel
        self.cdinv = np.linalg.inv(self.cd)

    @classmethod
    def from_header(cls, hdr):
        get = hdr.get if hasattr(hdr, 'get') else (lambda k, d=None: hdr[k] if k in hdr else d)
        c1, c2 = str(get('CTYPE1', '')).strip(), str(get('CTYPE2', '')).strip()
        if not (c1.startswith('RA--') and c1.endswith('TAN') and c2.startswith('DEC-') and c2.endswith('TAN')):
            raise ValueError(f'Error, only RA---TAN / DEC--TAN headers are supported (found {c1!r}, {c2!r}).')
        if get('A_ORDER') is not None or get('B_ORDER') is not None or get('PV1_1') is not None:
            raise ValueError('Error, distortion terms (SIP / PV) are not supported.')
        crpix = (float(get('CRPIX1')), float(get('CRPIX2')))
        crval = (float(get('CRVAL1')), float(get('CRVAL2')))
        if get('CD1_1') is not None:
            cd = [[float(get('CD1_1')), float(get('CD1_2', 0.0))], [float(get('CD2_1', 0.0)), float(get('CD2_2'))]]
        else:
            cdelt = (float(get('CDELT1')), float(get('CDELT2')))
            if get('PC1_1') is not None:
                pc = [[float(get('PC1_1')), float(get('PC1_2', 0.0))], [float(get('PC2_1', 0.0)), float(get('PC2_2', 1.0))]]
            else:
                rot = float(get('CROTA2', 0.0)) * D2R       # the AIPS convention
                pc = [[np.cos(rot), -np.sin(rot) * cdelt[1] / cdelt[0]], [np.sin(rot) * cdelt[0] / cdelt[1], np.cos(rot)]]
            cd = [[cdelt[0] * pc[0][0], cdelt[0] * pc[0][1]], [cdelt[1] * pc[1][0], cdelt[1] * pc[1][1]]]
        return cls(crpix, crval, cd)

    @classmethod
    def from_center(cls, ra, dec, pixscale_arcsec, shape):
        """North up, east left, `pixscale_arcsec` per pixel, (ra, dec) at the centre of a [ny, nx] image: the grid
        SWarp builds from -CENTER / -PIXEL_SCALE / -IMAGE_SIZE (resample_all.sh:334-338)."""
        ny, nx = shape
        s = pixscale_arcsec / 3600.0
        return cls(((nx + 1) / 2.0, (ny + 1) / 2.0), (ra, dec), [[-s, 0.0], [0.0, s]])

    def oversampled(self, n):
        """The same projection on a grid n times finer whose n x n blocks are this grid's pixels (FITS pixel centres are
        integers: CRPIX' = (CRPIX - 0.5) * n + 0.5)."""
        n = int(n)
        return type(self)(((self.crpix[0] - 0.5) * n + 0.5, (self.crpix[1] - 0.5) * n + 0.5), (self.crval[0], self.crval[1]),
                          [[self.cd[0, 0] / n, self.cd[0, 1] / n], [self.cd[1, 0] / n, self.cd[1, 1] / n]])

    def header_cards(self):
        return {'CTYPE1': 'RA---TAN', 'CTYPE2': 'DEC--TAN', 'CRPIX1': float(self.crpix[0]), 'CRPIX2': float(self.crpix[1]),
                'CRVAL1': float(self.crval[0]), 'CRVAL2': float(self.crval[1]), 'CD1_1': float(self.cd[0, 0]),
                'CD1_2': float(self.cd[0, 1]), 'CD2_1': float(self.cd[1, 0]), 'CD2_2': float(self.cd[1, 1])}

    def pix2sky(self, x, y):
        x, y = np.asarray(x, np.float64), np.asarray(y, np.float64)
        u, v = x + 1.0 - self.crpix[0], y + 1.0 - self.crpix[1]
        xi = (self.cd[0, 0] * u + self.cd[0, 1] * v) * D2R   # standard coordinates (radians), xi towards east
        eta = (self.cd[1, 0] * u + self.cd[1, 1] * v) * D2R
        a0, d0 = self.crval[0] * D2R, self.crval[1] * D2R
        den = np.cos(d0) - eta * np.sin(d0)
        ra = a0 + np.arctan2(xi, den)
        dec = np.arctan2((eta * np.cos(d0) + np.sin(d0)) * np.cos(ra - a0), den)
        return np.mod(ra / D2R, 360.0), dec / D2R

    def sky2pix(self, ra, dec):
        a, d = np.asarray(ra, np.float64) * D2R, np.asarray(dec, np.float64) * D2R
        a0, d0 = self.crval[0] * D2R, self.crval[1] * D2R
        cosc = np.sin(d0) * np.sin(d) + np.cos(d0) * np.cos(d) * np.cos(a - a0)
        with np.errstate(divide='ignore', invalid='ignore'):
            xi = np.where(cosc > 0, np.cos(d) * np.sin(a - a0) / cosc, np.nan) / D2R       # behind the tangent plane -> NaN
            eta = np.where(cosc > 0, (np.cos(d0) * np.sin(d) - np.sin(d0) * np.cos(d) * np.cos(a - a0)) / cosc, np.nan) / D2R
        u = self.cdinv[0, 0] * xi + self.cdinv[0, 1] * eta
        v = self.cdinv[1, 0] * xi + self.cdinv[1, 1] * eta
        return u + self.crpix[0] - 1.0, v + self.crpix[1] - 1.0


TILE_H, TILE_W = 16, 64          # the resample kernel's output tile (csrc/resample.hip)


def tile_affines(out_wcs, in_wcs, out_shape, tile_scale=1):
    """Per-tile affine approximation of the map output pixel -> input pixel (through the sky): the exact map at
    the tile centre plus its central-difference Jacobian over the tile.  Over a 64 x 16 pixel tile of an
    arcsecond-scale image the second-order terms of TAN -> TAN are < 1e-3 pixel (the kernel's phase table
    resolves 1/1024 pixel).  Returns float64 [tiles_y, tiles_x, 6]: xin = A0*x + A1*y + A2, yin = A3*x + A4*y + A5
    with ABSOLUTE output pixel coordinates x, y.  tile_scale = n: `out_wcs` / `out_shape` describe the n-times finer grid of
    an OVERSAMPLING n resample and a tile is the 64 n x 16 n fine pixels of one output tile (apgpu_resample_oversampled_f32)."""
    ny, nx = out_shape
    TILE_H, TILE_W = 16 * int(tile_scale), 64 * int(tile_scale)
    ty, tx = (ny + TILE_H - 1) // TILE_H, (nx + TILE_W - 1) // TILE_W
    yc = np.arange(ty, dtype=np.float64)[:, None] * TILE_H + (TILE_H - 1) / 2.0 + np.zeros((1, tx))
    xc = np.arange(tx, dtype=np.float64)[None, :] * TILE_W + (TILE_W - 1) / 2.0 + np.zeros((ty, 1))

    def fwd(x, y):
        ra, dec = out_wcs.pix2sky(x, y)
        return in_wcs.sky2pix(ra, dec)

    hx, hy = TILE_W / 2.0, TILE_H / 2.0
    x0, y0 = fwd(xc, yc)
    xpx, ypx = fwd(xc + hx, yc)
    xmx, ymx = fwd(xc - hx, yc)
    xpy, ypy = fwd(xc, yc + hy)
    xmy, ymy = fwd(xc, yc - hy)
    a0, a3 = (xpx - xmx) / (2 * hx), (ypx - ymx) / (2 * hx)
    a1, a4 = (xpy - xmy) / (2 * hy), (ypy - ymy) / (2 * hy)
    a2 = x0 - a0 * xc - a1 * yc
    a5 = y0 - a3 * xc - a4 * yc
    return np.ascontiguousarray(np.stack([a0, a1, a2, a3, a4, a5], axis=-1))
