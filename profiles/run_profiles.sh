#!/bin/bash
# Collects the rocprofv3 evidence for bench.py on the GPU box (run through gpurun from the repo root):
#   1. --kernel-trace --stats  : per-kernel average durations (must agree with bench.py's HIP-event timings)
#   2. --pmc FETCH_SIZE        : HBM read traffic  (separate pass, TCC slots: MI355X_MICROARCH.md "rocprofv3 PMC slots")
#   3. --pmc WRITE_SIZE        : HBM write traffic (separate pass)
# Outputs land in gpurun_out/prof_<tag>/ ; copy the summaries into profiles/ afterwards.
set -u
TAG=${1:-r01}
ARGS=${2:-"--steps 5 --warmup 2 --no-cpu-baseline"}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o bench -- python3 $ROOT/bench.py $ARGS > $OUT/bench_trace.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o bench -- python3 $ROOT/bench.py $ARGS > $OUT/bench_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o bench -- python3 $ROOT/bench.py $ARGS > $OUT/bench_write.log 2>&1
find $OUT -name "*.csv" | head -20
