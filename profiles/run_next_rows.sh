#!/bin/bash
# rocprofv3 kernel-trace of profiles/next_rows_bench.py (kernel-only durations of the 8(f) rows).
set -u
TAG=${1:-r01}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_${TAG}_next
mkdir -p $OUT
python3 $ROOT/profiles/next_rows_bench.py > $OUT/next_rows.jsonl 2> $OUT/next_rows.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o next -- python3 $ROOT/profiles/next_rows_bench.py > $OUT/next_rows_under_rocprof.jsonl 2>&1
cat $OUT/next_rows.jsonl
grep -E "psdev" $OUT/trace/next_kernel_stats.csv | cut -c1-60,150-400 | head -12
