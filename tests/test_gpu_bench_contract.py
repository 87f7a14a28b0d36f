"""bench.py contract: one JSON line on stdout with the agreed keys (small sizes, so this takes seconds)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(*extra):
    cmd = [sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '1', '--steps', '3', '--warmup', '1', '--height', '256', '--width', '512',
           '--cpu-seconds', '0.5', *extra]
    r = subprocess.run(cmd, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout                          # exactly one line, and it is the JSON
    return json.loads(lines[0])


def test_default_workload_line():
    d = _run('--frames', '16')
    for k in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling', 'vs_baseline',
              'dtype', 'data', 'config', 'roofline', 'cpu_baseline'):
        assert k in d, k
    assert d['n_gpus'] == 1 and d['steps'] == 3 and d['warmup'] == 1 and d['higher_is_better'] is True and d['vs_baseline'] is None
    assert d['unit'] == 'Mpixels/s' and d['dtype'] == 'f32' and d['data'] == 'synthetic' and 'workload' in d['config']
    rf = d['roofline']
    assert rf['bound'] == 'hbm' and rf['unit'] == 'GB/s' and rf['peak'] == 8000.0 and abs(rf['frac'] - rf['achieved'] / rf['peak']) < 1e-12
    assert rf['algorithmic_bytes'] == 4 * 16 * 256 * 512 + 16 * 256 * 512 and rf['avg_launch_ms'] > 0
    assert rf['kernel'] == 'stack_fast_kernel<16, float, true, true, 0, false>'      # reported by the library's dispatch (the fast kernel + redo list)
    assert d['rccl_world_size'] == 1 and len(d['per_rank_ms']) == 1
    cb = d['cpu_baseline']
    assert cb['kind'] == 'port' and cb['cores'] >= 1 and cb['value'] > 0 and 'sample' in cb
    assert cb['numpy']['single_process']['value'] > 0 and cb['numpy']['single_process']['cores'] == 1 and cb['numpy']['cpu']
    assert abs(d['value'] - 16 * 256 * 512 / 1e6 / (d['ms_per_step'] * 1e-3)) / d['value'] < 1e-6


def test_other_workloads_run():
    d = _run('--workload', 'c4', '--frames', '16', '--no-cpu-baseline')
    assert d['dtype'] == 'u16' and 'median' in d['metric'] and 'cpu_baseline' not in d
    d = _run('--workload', 'c5', '--frames', '4', '--no-cpu-baseline')
    assert 'resample' in d['metric']
    for ex in ('f64', 'f32'):
        d = _run('--frames', '16', '--force-collective', '--no-cpu-baseline', '--stripes', '3', '--exchange', ex)
        assert d['n_gpus'] == 1 and d['value'] > 0 and d['exchange_bytes_per_pixel'] == (16 if ex == 'f64' else 8) and d['stripes'] == 3 and d['exchange_ms'] is not None
    d = _run('--frames', '16', '--parallelism', 'rowshard', '--no-cpu-baseline')
    assert d['n_gpus'] == 1 and d['value'] > 0


def test_gpus_flag_must_match_the_world():
    cmd = [sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '1', '--no-cpu-baseline']
    env = dict(os.environ, WORLD_SIZE='1', RANK='0', LOCAL_RANK='0')
    r = subprocess.run(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
    assert r.returncode != 0 and not r.stdout.strip()


def test_multi_rank_paths_on_one_gpu():
    """The N > 1 code of bench.py with REAL kernels on a one-GPU box: APGPU_BENCH_ONE_GPU_TEST puts every rank on GPU 0 and the
    ranks on gloo, so the built-in launcher, N-sharding, the striped one-all-reduce-per-stripe exchange, the hierarchical
    shards of --scaling strong, the row-shard leg and every reported field run end to end (the numbers mean nothing)."""
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_PORT', 'MASTER_ADDR')}
    env['APGPU_BENCH_ONE_GPU_TEST'] = '1'
    base = [sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '2', '--warmup', '1', '--height', '128',
            '--width', '512', '--frames', '16', '--no-cpu-baseline']
    for extra in ([], ['--exchange', 'f32'], ['--exchange', 'f64'], ['--scaling', 'strong', '--total-frames', '64', '--hier-shards', '4'],
                  ['--parallelism', 'rowshard'], ['--scaling', 'strong', '--total-frames', '64', '--parallelism', 'rowshard']):
        r = subprocess.run(base + extra, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
        assert r.returncode == 0, (extra, r.stderr[-3000:])
        lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
        assert len(lines) == 1, r.stdout
        d = json.loads(lines[0])
        assert d['n_gpus'] == 2 and d['rccl_world_size'] == 2 and len(d['per_rank_ms']) == 2 and d['value'] > 0 and d['one_gpu_test']
        if '--parallelism' not in extra:
            assert d['exchange_ms'] is not None and d['exchange_ms'] >= 0 and d['stripes'] >= 1
            assert d['rowshard']['value'] > 0 and d['rowshard']['ms_per_step'] > 0
            assert d['exchange'] == (extra[1] if '--exchange' in extra else 'rs')
            # 'rs': float64 sum + float16 count (the job has at most 2048 frames) reduce-scattered, float32 mean rows all-gathered
            assert d['exchange_bytes_per_pixel'] == {'rs': 10, 'f64': 16, 'f32': 8}[d['exchange']]
            assert d.get('exchange_count_dtype') == ('float16' if d['exchange'] == 'rs' else None)
            px = 128 * 512
            assert d['exchange_bytes_on_wire'] == {'rs': int(0.5 * 14 * px), 'f64': int(2 * 0.5 * 16 * px), 'f32': int(2 * 0.5 * 8 * px)}[d['exchange']]
        if '--scaling' in extra and '--parallelism' not in extra:
            assert d['hier_shards'] == 4 and d['config']['frames_per_gpu'] == 32 and d['config']['frames_total'] == 64
            assert d['roofline']['kernel'].startswith('stack_fast_kernel<16,')        # shards of 16 frames
