#!/bin/bash
# One SQ-counter pass (separate from any trace pass) of the single-chain bench step for a given library build:
#   r04_sq_pass.sh <tag> [path to lib .so]
TAG=$1; LIB=${2:-}
export GPU_MAX_HW_QUEUES=8
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
[ -n "$LIB" ] && export PUTSLAM_HIP_LIB=$ROOT/$LIB
OUT=$ROOT/gpurun_out/prof_$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE \
  --output-format csv -d $OUT/sq -o bench -- python3 $ROOT/bench.py --streams 1 --steps 20 --warmup 20 --warm-seconds 0 --repeats 1 --no-other-modes --no-cpu-baseline > $OUT/bench_sq.log 2>&1
