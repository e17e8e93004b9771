#!/bin/bash
# Collects the rocprofv3 evidence for bench.py on the GPU box (run through gpurun from the repo root):
#   1. --kernel-trace --stats  : per-kernel average durations (must agree with bench.py's HIP-event timings)
#   2. --pmc FETCH_SIZE        : HBM read traffic  (separate pass, TCC slots: MI355X_MICROARCH.md "rocprofv3 PMC slots")
#   3. --pmc WRITE_SIZE        : HBM write traffic (separate pass)
# Outputs land in gpurun_out/prof_<tag>/ ; copy the summaries into profiles/ afterwards.
# The profiled command is the single launch chain (--streams 1): every kernel has the chip to itself, which is what
# bench.py's `roofline` / `kernel_ms` (its single-chain leg) report; the default (two chains out of step) timing of `value` is in
# bench_default.json of the same directory.
set -u
TAG=${1:-r02}
# 20 warm-up steps: the chip needs ~10 launches of this step to reach its steady clock (r02g/kernel_launch_ms.json),
# and rocprofv3's --stats average runs over every launch of the process
ARGS=${2:-"--streams 1 --steps 20 --warmup 20 --warm-seconds 0 --repeats 1 --no-other-modes --no-cpu-baseline"}
# bench.py sets this with os.environ.setdefault, but under rocprofv3 the profiler's preloaded library may initialise the
# HIP runtime before Python runs: export it here so that profiled and un-profiled runs use the same queue count
export GPU_MAX_HW_QUEUES=16
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o bench -- python3 $ROOT/bench.py $ARGS > $OUT/bench_trace.log 2>&1
timeout 900 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o bench -- python3 $ROOT/bench.py $ARGS > $OUT/bench_fetch.log 2>&1
timeout 900 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o bench -- python3 $ROOT/bench.py $ARGS > $OUT/bench_write.log 2>&1
# SQ / matrix-pipe counters of the same command (own pass): VALU and MFMA instruction counts, busy cycles
timeout 900 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE \
  --output-format csv -d $OUT/sq -o bench -- python3 $ROOT/bench.py $ARGS > $OUT/bench_sq.log 2>&1
cd $ROOT && python3 bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
find $OUT -name "*.csv" | head -20
