"""bench.py's built-in launcher: `python bench.py --gpus N` must start N ranks itself, rendezvous, and rank 0 prints
ONE JSON line; a world size that differs from --gpus is an error.  Driven on CPU with --selftest-cpu (gloo, no kernels);
the same launcher path with RCCL and the real kernels runs in tests/test_gpu_multi.py when two GPUs are present."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, 'bench.py')


def _env():
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_PORT', 'MASTER_ADDR')}
    return env


@pytest.mark.timeout(300)
def test_launcher_spawns_two_ranks():
    r = subprocess.run([sys.executable, BENCH, '--gpus', '2', '--steps', '3', '--selftest-cpu'], env=_env(),
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=280)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    lines = [ln for ln in r.stdout.decode().splitlines() if ln.strip()]
    assert len(lines) == 1, lines                                     # exactly one JSON line, from rank 0
    d = json.loads(lines[0])
    assert d['n_gpus'] == 2 and d['rccl_world_size'] == 2 and len(d['per_rank_ms']) == 2 and d['selftest'] is True


@pytest.mark.timeout(300)
def test_world_size_mismatch_is_an_error():
    env = _env()
    env.update(WORLD_SIZE='1', RANK='0', LOCAL_RANK='0')              # "launched" as one rank but asked for two
    r = subprocess.run([sys.executable, BENCH, '--gpus', '2', '--selftest-cpu'], env=env, stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, timeout=280)
    assert r.returncode != 0
    assert b'WORLD_SIZE=1' in r.stderr and not r.stdout.strip()


@pytest.mark.timeout(300)
def test_single_rank_selftest():
    r = subprocess.run([sys.executable, BENCH, '--selftest-cpu', '--steps', '2'], env=_env(), stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, timeout=280)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    d = json.loads(r.stdout.decode().strip())
    assert d['n_gpus'] == 1 and d['rccl_world_size'] == 1
