#!/usr/bin/env python3
"""Time-budgeted randomised parity run (development aid): random stacks (1 .. 512 frames, float32 / uint16, fused calibration
with awkward masters, pixel masks, every clip option and output plane) and the test suite's own random cases with fresh seeds,
HIP kernels against the oracle.  Prints every mismatch and a summary.

    python tools/fuzz_long.py --minutes 8 [--seed0 100000]
"""
import argparse
import os
import sys
import time
import traceback

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from astrophotography_amd import ops
from oracle import apref
from tests import test_gpu_fuzz as tf
from tests.util import assert_ulp, synth_cube, synth_masters


def dev(a):
    a = np.ascontiguousarray(a)
    return ops.to_device_u16(a) if a.dtype == np.uint16 else torch.from_numpy(a).cuda()


def big_case(seed):
    return tf.wide_case(ops, apref, seed)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--minutes', type=float, default=5.0)
    ap.add_argument('--seed0', type=int, default=100000)
    a = ap.parse_args()
    t_end = time.time() + 60.0 * a.minutes
    fails, runs = [], {'big': 0, 'stack': 0, 'image': 0}
    seed = a.seed0
    while time.time() < t_end:
        for name, fn in (('big', lambda s: big_case(s)), ('stack', lambda s: tf.test_random_stack_configs(ops, apref, s)),
                         ('image', lambda s: tf.test_random_image_kernels(ops, apref, s))):
            try:
                fn(seed)
            except Exception as ex:                       # noqa: BLE001 - a fuzz driver reports everything
                fails.append((name, seed, str(ex)[:600]))
                print('FAIL', name, seed, str(ex)[:600], flush=True)
                if not isinstance(ex, AssertionError):
                    traceback.print_exc()
            runs[name] += 1
        seed += 1
    print('runs', runs, 'failures', len(fails))
    return 1 if fails else 0


if __name__ == '__main__':
    raise SystemExit(main())
