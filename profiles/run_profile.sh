#!/bin/bash
# Collects the rocprofv3 evidence for bench.py's dominant kernel on the GPU box (run through gpurun):
#   bash profiles/run_profile.sh <tag>
# 1. --kernel-trace --stats  -> per-kernel durations (must agree with bench.py's HIP-event timing)
# 2. --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes -> HBM traffic (MI355X_MICROARCH.md: on
#    gfx950 FETCH_SIZE reports half of the bytes of a wide coalesced read stream; units are KiB).
TAG=${1:-r01}
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $REPO/bench.py --steps 10 --warmup 2 --no-cpu-baseline > $OUT/bench_trace.json 2> $OUT/trace.log
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $REPO/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/bench_fetch.json 2> $OUT/fetch.log
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $REPO/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/bench_write.json 2> $OUT/write.log
find $OUT -name "*.csv" | head -20
python3 $REPO/profiles/summarize.py $OUT $TAG
