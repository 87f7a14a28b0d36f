#!/bin/bash
# Collects the rocprofv3 evidence for bench.py's dominant kernel on the GPU box (run through gpurun):
#   bash profiles/run_profile.sh <tag> [bench.py arguments ...]
# ONE invocation = one box = one consistent set: every pass runs the same `python3 bench.py ...` command
# (program directly after `--`), and profiles/summarize.py condenses all of them into a single summary.
#  1. --kernel-trace --stats        -> per-kernel durations (must agree with bench.py's HIP-event timing)
#  2. --pmc FETCH_SIZE              -> HBM read traffic   (separate passes: MI355X_MICROARCH.md - the TCC block
#  3. --pmc WRITE_SIZE              -> HBM write traffic   cannot hold both; FETCH_SIZE is doubled on gfx950; KiB units)
#  4. --pmc SQ_INSTS_VALU SQ_WAVES  -> VALU instructions per wave (the VALU-issue-bound claim of DESIGN 4.1)
#  5. --pmc SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS
#  6. --pmc GRBM_GUI_ACTIVE         -> busy cycles of the dispatch -> the clock the chip actually held
#  7. --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU -> average resident waves, VALU busy fraction
TAG=${1:-r02}
shift
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/prof_$TAG
rm -rf $OUT
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
python3 $REPO/bench.py --steps 20 --warmup 3 "$@" > $OUT/bench_line.json 2> $OUT/bench_line.log
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $REPO/bench.py --steps 10 --warmup 2 --no-cpu-baseline "$@" > $OUT/bench_trace.json 2> $OUT/trace.log
i=0
for ctr in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_WAVES" "SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS" "GRBM_GUI_ACTIVE" "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $ctr --output-format csv -d $OUT/pmc_$i -- python3 $REPO/bench.py --steps 3 --warmup 1 --no-cpu-baseline "$@" > $OUT/bench_pmc_$i.json 2> $OUT/pmc_$i.log
done
python3 $REPO/profiles/summarize.py $OUT $TAG
# the raw rocprofv3 output is large (gpurun merges at most 64 MiB back): keep the summaries only
rm -rf $OUT/trace $OUT/pmc_[0-9]*
