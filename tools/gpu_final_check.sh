#!/bin/bash
# final check of a round on one box: build() is a no-op there (the .so travels), smoke(), the GPU suite, the default bench line
cd ${GRAFT_REPO_ROOT:-/root/repo}
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep "smoke"
python -m pytest tests -m gpu -q 2>&1 | grep -E "passed|failed" | tail -2
python bench.py 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('C2', round(d['ms_per_step'],4), 'ms frac', round(r['frac'],4), 'valu', r['valu'], 'traffic', r['traffic'], 'cpu', d['cpu_baseline']['value'], d['cpu_baseline']['cores'])"
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --steps 5 --warmup 2 2>/dev/null | tail -1 | cut -c1-160
