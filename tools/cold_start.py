#!/usr/bin/env python3
"""Cold start of a fresh process (the reference's flow is one process per frame, calibrate_all.sh:406-411):
wall time from interpreter start to the first result of each hot-path entry point, split into its phases.

    python tools/cold_start.py [--repeat 3]        (every repetition is a NEW process)
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r'''
import json, os, sys, time
t0 = time.perf_counter()
sys.path.insert(0, %r)
import torch
t1 = time.perf_counter()
torch.cuda.init(); torch.zeros(1, device='cuda'); torch.cuda.synchronize()
t2 = time.perf_counter()
from astrophotography_amd import _lib, ops, synth
lib = _lib.load()
t3 = time.perf_counter()
dev = torch.device('cuda', 0)
H, W, N = 1024, 1024, 16
masters = synth.make_masters(H, W, config_id=2, device=dev)
nflat, _ = ops.flat_normalize(masters['flat'])
torch.cuda.synchronize()
t4 = time.perf_counter()
frames = synth.make_frames(N, masters, nflat, config_id=2, dtype=torch.float32, first_frame=0)
torch.cuda.synchronize()
t5 = time.perf_counter()
cal = ops.calibrate(frames[0], masters['bias'], masters['dark'], nflat, synth.EXP_RATIO)
torch.cuda.synchronize()
t6 = time.perf_counter()
calib = dict(bias=masters['bias'], dark=masters['dark'], nflat=nflat, exp_ratio=synth.EXP_RATIO)
r = ops.stack_sigclip(frames, sigma=3.0, maxiters=5, calib=calib, outputs=('mean',))
torch.cuda.synchronize()
t7 = time.perf_counter()
r = ops.stack_sigclip(frames, sigma=3.0, maxiters=5, calib=calib, outputs=('mean',))
torch.cuda.synchronize()
t8 = time.perf_counter()
print(json.dumps(dict(import_torch=t1 - t0, gpu_init=t2 - t1, load_libapgpu=t3 - t2, first_flat_normalize=t4 - t3, synth_frames=t5 - t4,
                      first_calibrate=t6 - t5, first_stack=t7 - t6, second_stack=t8 - t7, library_bytes=os.path.getsize(_lib.LIB_PATH))))
'''


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--repeat', type=int, default=3)
    args = ap.parse_args()
    rows = []
    for k in range(args.repeat):
        t0 = time.perf_counter()
        r = subprocess.run([sys.executable, '-c', CHILD % ROOT], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
        wall = time.perf_counter() - t0
        if r.returncode != 0:
            print(r.stderr[-2000:], file=sys.stderr)
            return 1
        d = json.loads(r.stdout.strip().splitlines()[-1])
        d['process_wall'] = wall
        rows.append(d)
    keys = ['import_torch', 'gpu_init', 'load_libapgpu', 'first_flat_normalize', 'first_calibrate', 'first_stack', 'second_stack', 'process_wall']
    print('# cold start of a fresh process, seconds (library: %.1f MB); run 1 pages the image in, the later ones are warm-cache starts' % (rows[0]['library_bytes'] / 1e6))
    print('# ' + ' '.join('%20s' % k for k in keys))
    for d in rows:
        print('  ' + ' '.join('%20.4f' % d[k] for k in keys))
    return 0


if __name__ == '__main__':
    sys.exit(main())
