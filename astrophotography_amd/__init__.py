"""astrophotography_amd - MI355X-native calibrate-and-stack hot path of AstroPhotography.

Drop-in names for the reference's hot-path API (``import AstroPhotography as ap``; reference
``AstroPhotography/__init__.py:10-12``, ``core/__init__.py:6-19``): ``ApCalibrate``, ``ApFindBadPixels``,
``ApFixBadPixels``, ``ApImArith``, ``ApMasterCal`` plus the new slab-level ``ApStack``/``ApCombine``.
Names resolve lazily so that importing the package needs neither torch nor the HIP library.
"""
__version__ = '0.1.0'

_LAZY = {
    'ApCalibrate': ('.core.ApCalibrate', 'ApCalibrate'),
    'ApFindBadPixels': ('.core.ApFindBadPixels', 'ApFindBadPixels'),
    'ApFixBadPixels': ('.core.ApFixBadPixels', 'ApFixBadPixels'),
    'ApImArith': ('.core.ApImArith', 'ApImArith'),
    'ApMasterCal': ('.core.ApMasterCal', 'ApMasterCal'),
    'ApStack': ('.core.ApStack', 'ApStack'),
    'ApCombine': ('.core.ApStack', 'ApCombine'),
    'ApResample': ('.core.ApResample', 'ApResample'),
    'ApImageDifference': ('.core.ApCalcReadNoise', 'ApImageDifference'),
    'ApCalcReadNoise': ('.core.ApCalcReadNoise', 'ApCalcReadNoise'),
    'ApMeasureBackground': ('.core.ApMeasureBackground', 'ApMeasureBackground'),
    'ApFixCosmicRays': ('.core.ApFixCosmicRays', 'ApFixCosmicRays'),
}

__all__ = sorted(_LAZY) + ['__version__']


def __getattr__(name):
    if name in _LAZY:
        import importlib
        mod, attr = _LAZY[name]
        return getattr(importlib.import_module(mod, __name__), attr)
    raise AttributeError('module %r has no attribute %r' % (__name__, name))
