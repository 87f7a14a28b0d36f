#!/bin/bash
# one-off measurement script of round 4 (pixel-granular redo list): run through gpurun
mkdir -p gpurun_out/r04i
python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|Error|FAILED" | tail -5
export APGPU_LIBRARY=$PWD/build_variants/pix/libapgpu.so
python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_classes.py -x -q 2>&1 | grep -E "passed|failed|Error|FAILED|assert" | tail -8
unset APGPU_LIBRARY
bash tools/ab_variants.sh 4 prod pix 2>&1 | tail -2
AB_ARGS="--workload c5" bash tools/ab_variants.sh 2 prod pix 2>&1 | tail -2
APGPU_DEBUG_REDO=1 APGPU_LIBRARY=$PWD/build_variants/pixdev/libapgpu.so python bench.py --no-cpu-baseline --steps 2 --warmup 1 2>&1 | grep stack_fast | head -1
APGPU_DEBUG_REDO=1 APGPU_LIBRARY=$PWD/build_variants/pixdev/libapgpu.so python bench.py --workload c5 --no-cpu-baseline --steps 2 --warmup 1 2>&1 | grep stack_fast | head -1
cd /tmp && export TMPDIR=/tmp
export APGPU_LIBRARY=$GRAFT_REPO_ROOT/build_variants/pix/libapgpu.so
for wl in c2 c5; do
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/tr_$wl -- python3 $GRAFT_REPO_ROOT/bench.py --workload $wl --steps 10 --warmup 2 --no-cpu-baseline > /dev/null 2>&1
  f=$(find /tmp/tr_$wl -name "*kernel_stats.csv" | head -1)
  echo "== $wl"; head -8 $f | cut -c1-200
done
