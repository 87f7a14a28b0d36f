#!/usr/bin/env python3
"""Kernel-trace target for the global sigma clip (development aid): rocprofv3 --kernel-trace --stats -- python3 tools/prof_global.py"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from astrophotography_amd import ops, synth

masters = synth.make_masters(4096, 4096, config_id=2, device='cuda')
for _ in range(5):
    r = ops.sigclip_global(masters['dark'], sigma=4.0, maxiters=5)
torch.cuda.synchronize()
print(r)
