import json
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)


def meta(g, key):
    return json.loads(str(g[key]))


def f32_ordered(a):
    """float32 -> int64 keys with the same ordering (for ulp distances)."""
    b = np.ascontiguousarray(a, dtype=np.float32).view(np.int32).astype(np.int64)
    return np.where(b < 0, -(b & 0x7fffffff), b)


def ulp_diff(a, b):
    """max |ulp distance| between float32 arrays; NaN must match NaN exactly in position."""
    a = np.asarray(a, np.float32)
    b = np.asarray(b, np.float32)
    na, nb = np.isnan(a), np.isnan(b)
    assert np.array_equal(na, nb), 'NaN positions differ: %d vs %d' % (na.sum(), nb.sum())
    d = np.abs(f32_ordered(a) - f32_ordered(b))
    d[na] = 0
    return d


def assert_ulp(a, b, tol=1, what=''):
    d = ulp_diff(a, b)
    assert d.max(initial=0) <= tol, '%s: max ulp diff %d (>%d) at %s; %d of %d differ' % (
        what, d.max(), tol, np.unravel_index(d.argmax(), d.shape), (d > 0).sum(), d.size)
    return float((d == 0).mean()) if d.size else 1.0


def assert_biteq(a, b, what=''):
    a = np.ascontiguousarray(a)
    b = np.ascontiguousarray(b)
    assert a.dtype == b.dtype and a.shape == b.shape, (what, a.dtype, b.dtype, a.shape, b.shape)
    if a.dtype.kind == 'f':
        v = {4: np.uint32, 8: np.uint64}[a.dtype.itemsize]
        ok = (a.view(v) == b.view(v)) | (np.isnan(a) & np.isnan(b))
    else:
        ok = a == b
    assert ok.all(), '%s: %d of %d differ, first at %s' % (what, (~ok).sum(), ok.size, np.argwhere(~ok)[:3].tolist())


def synth_cube(rng, N, shape, nan_frac=0.0, dtype=np.float32):
    """Frames with Gaussian noise, cosmic-ray-like positive outliers and a few low outliers."""
    cube = rng.normal(500, 20, (N,) + tuple(shape))
    hits = rng.random(cube.shape) < 0.02
    cube[hits] += rng.uniform(100, 5000, hits.sum())
    lows = rng.random(cube.shape) < 0.005
    cube[lows] -= rng.uniform(100, 400, lows.sum())
    if dtype == np.uint16:
        return np.clip(np.rint(cube), 0, 65535).astype(np.uint16)
    cube = cube.astype(np.float32)
    if nan_frac > 0:
        bad = rng.random(cube.shape) < nan_frac
        vals = rng.choice(np.array([np.nan, np.inf, -np.inf], np.float32), bad.sum())
        cube[bad] = vals
    return cube


def synth_masters(rng, shape):
    bias = rng.normal(1000, 5, shape).astype(np.float32)
    dark = rng.normal(20, 3, shape).astype(np.float32)
    hot = rng.random(shape) < 0.002
    dark[hot] = rng.uniform(2000, 6000, hot.sum()).astype(np.float32)
    flat = rng.normal(30000, 300, shape).astype(np.float32)
    return bias, dark, flat
