#!/bin/bash
# rocprofv3 evidence for BASELINE configs[4] (8 pairs x 5000 keypoints x H = 100 000, fixed schedule; the loop
# of reference src/TransformEst/RANSAC.cpp:87-150) at the current library: for each error version a kernel trace + stats pass
# and the two HBM traffic passes (FETCH_SIZE / WRITE_SIZE, separate, as MI355X_MICROARCH.md prescribes).  One launch chain,
# Python submission (every kernel alone on the chip).  usage (GPU box, repo root): bash profiles/run_stress_profiles.sh r06t
# Then, here: python profiles/summarize.py <tag>_E0 9x5000xH100000xE0xfixedxmfmaxfast ; the same for E1.
set -u
TAG=${1:-r06t}
export GPU_MAX_HW_QUEUES=16
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
for E in 0 1; do
  OUT=$ROOT/gpurun_out/prof_${TAG}_E$E
  mkdir -p $OUT
  ARGS="--preset stress --error-version $E --streams 1 --submit python --steps 10 --warmup 10 --warm-seconds 0 --repeats 1 --no-other-modes --no-cpu-baseline"
  cd /tmp && export TMPDIR=/tmp
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o bench -- python3 $ROOT/bench.py $ARGS > $OUT/bench_trace.log 2>&1
  timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o bench -- python3 $ROOT/bench.py $ARGS > $OUT/bench_fetch.log 2>&1
  timeout 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o bench -- python3 $ROOT/bench.py $ARGS > $OUT/bench_write.log 2>&1
  cd $ROOT
done
find $ROOT/gpurun_out -path "*prof_${TAG}_E*" -name "*.csv" | head -20
