#!/bin/bash
python -m pytest tests/test_gpu_redo.py tests/test_gpu_parity.py tests/test_gpu_classes.py tests/test_gpu_f64.py -m gpu -x -q 2>&1 | tail -8
python - <<'PY'
import torch, numpy as np
from astrophotography_amd import ops
def t(fn, n=5):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n
for N in (96, 128):
    for dt in ('f32', 'u16'):
        g = torch.Generator(device='cuda').manual_seed(3)
        fr = (torch.randn((N, 4096, 4096), device='cuda', generator=g) * 12 + 1000)
        if dt == 'u16':
            fr = fr.round_().clamp_(0, 65535).to(torch.int32).to(torch.uint16)
        kw = dict(sigma=5.0, maxiters=1, cenfunc='median', stdfunc='mad_std', outputs=('mean_f64', 'std_f64', 'count'))
        fast = t(lambda: ops.stack_sigclip(fr, **kw))
        rich = t(lambda: ops.stack_sigclip(fr, single_kernel=True, **kw), 2)
        a = ops.stack_sigclip(fr, **kw); b = ops.stack_sigclip(fr, single_kernel=True, **kw)
        print('A6 %d x 4096^2 %s: fast pair %.3f ms, rich kernel alone %.3f ms; counts equal %s, max |dmean| %.2e' % (
            N, dt, fast, rich, bool(torch.equal(a['count'], b['count'])), float((a['mean_f64'] - b['mean_f64']).abs().max())))
        del fr
PY
