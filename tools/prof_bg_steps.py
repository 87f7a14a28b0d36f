#!/usr/bin/env python3
"""Wall time of the steps of ApMeasureBackground.process_data on a 4096^2 frame (development aid)."""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from astrophotography_amd import ops, synth
from astrophotography_amd.core import ApMeasureBackground as M

cal, _ = synth.make_sky_frame(4096, 4096)
mb = M.ApMeasureBackground('ERROR')


def t(name, fn, reps=3):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        r = fn()
    torch.cuda.synchronize()
    print('%-34s %8.3f ms' % (name, 1e3 * (time.perf_counter() - t0) / reps), flush=True)
    return r


t('process_data (all)', lambda: mb.process_data(cal))
mask = t('_make_source_mask', lambda: mb._make_source_mask(cal))
t('  sigclip_global maxiters 10', lambda: ops.sigclip_global(cal, sigma=3.0, maxiters=10))
mb._set_bgbox_size(4096, 4096, None, None, None, None)
bh, bw = mb._boxsize
st = t('box_clipped_stats + .cpu()', lambda: ops.box_clipped_stats(cal, mask, bh, bw, sigma=3.0, maxiters=5).cpu().numpy())
med = st[..., 0]
good = np.ones(med.shape, bool)
good[3, 4] = False
mesh = np.where(good, med, np.nan)
t('_fill_excluded', lambda: M._fill_excluded(mesh, good))
mesh = M._fill_excluded(mesh, good)
t('_nanmedian_filter', lambda: M._nanmedian_filter(mesh, 3))
t('_bspline3_prefilter', lambda: M._bspline3_prefilter(mesh))
coef = torch.from_numpy(M._bspline3_prefilter(mesh)).cuda()
t('spline_zoom', lambda: ops.spline_zoom(coef, bh, bw, 4096, 4096, float(mesh.min()), float(mesh.max())))
