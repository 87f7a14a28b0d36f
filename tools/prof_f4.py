#!/usr/bin/env python3
"""Kernel-trace target for the per-frame F4 steps (development aid): background mesh and cosmic rays on a 4096^2 frame."""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from astrophotography_amd import ops, synth
from astrophotography_amd.core.ApMeasureBackground import ApMeasureBackground

cal, _ = synth.make_sky_frame(4096, 4096)
mb = ApMeasureBackground('ERROR')
what = sys.argv[1] if len(sys.argv) > 1 else 'both'
for _ in range(3):
    if what in ('both', 'bg'):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        mb.process_data(cal)
        torch.cuda.synchronize(); print('background wall ms', 1e3 * (time.perf_counter() - t0))
    if what in ('both', 'cr'):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        r = ops.lacosmic(cal, None, niter=4)
        torch.cuda.synchronize(); print('lacosmic wall ms', 1e3 * (time.perf_counter() - t0), 'iterations', r[2])
