"""Development aid: which share of 16-frame columns leaves the float32 fast path (listed for the redo pass), by the number of
NaN values in the column - same frames for every pixel, or frames drawn per pixel - and by noise level?"""
import sys
import numpy as np
import torch
sys.path.insert(0, '.')
from astrophotography_amd import ops

N, H, W = 16, 256, 1024
rng = np.random.default_rng(1)
for noise in (6.0, 60.0):
    for per_pixel in (False, True):
        for k in range(0, 7):
            cube = rng.normal(500.0, noise, (N, H, W)).astype(np.float32)
            if per_pixel:
                order = rng.random((N, H, W)).argsort(axis=0)          # a random permutation of the frames per pixel
                bad = order < k
                cube[bad] = np.nan
            else:
                for j in range(k):
                    cube[(5 + 2 * j) % N] = np.nan
            d = torch.from_numpy(cube).cuda()
            ops.stack_redo_stats(reset=True)
            ops.stack_sigclip(d, sigma=3.0, maxiters=5, outputs=('mean',))
            torch.cuda.synchronize()
            st = ops.stack_redo_stats()
            print('noise %4.0f  NaNs per column %d (%s): listed %.4f of the pixels, blocks given up %d of %d' % (
                noise, k, 'drawn per pixel' if per_pixel else 'same frames', st['pixels_listed'] / st['pixels'], st['blocks_given_up'], H * W // 64))
