#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
python bench.py --no-cpu-baseline --steps 20 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('C2', d['ms_per_step'], d['roofline']['frac'], d['redo_fraction'])"
python bench.py --workload c5 --no-cpu-baseline --steps 10 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('C5', d['ms_per_step'], d['redo_fraction'])"
python tools/bench_fused.py 2>&1 | grep "N=\|equal" | tail -2
