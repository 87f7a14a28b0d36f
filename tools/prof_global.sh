#!/bin/bash
# Per-kernel times of the global sigma clip (development aid): gpurun -- 'bash tools/prof_global.sh'
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$REPO/gpurun_out/prof_global
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/t -- python3 $REPO/tools/prof_global.py > $OUT/run.log 2>&1
f=$(find $OUT/t -name '*kernel_stats.csv' | head -1)
cut -c1-150 $f | head -30
