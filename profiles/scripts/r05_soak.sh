#!/bin/bash
# Soak of the round-5 library: single-pair configurations with EVERY hypothesis's count compared (all error versions and
# schedules), random batches through the staged scoring (reordered / original order / complete / oracle; the keys block checked
# all-ones at rest every 8th batch), and the pipelined stream against the batched call.
# usage: r05_soak.sh <out dir> [single iters] [reprojection iters] [batch iters] [seed base]
# Exit code: the worst of the fuzz commands; the per-worker logs of a FAILING command are kept in full (round
# 4's script truncated every log and always exited 0).
OUT=${1:-gpurun_out/r05_soak}; mkdir -p $OUT/logs
S=${5:-606060}
worst=0
run() { # name, log tag, fuzz arguments...
  name=$1; tag=$2; shift 2
  python3 tests/fuzz_gpu.py "$@" --log-dir $OUT/logs --tag $tag > $OUT/$name.txt 2>&1
  rc=$?
  tail -n 2 $OUT/$name.txt
  if [ $rc -ne 0 ]; then
    echo "$name: exit code $rc -- full worker logs kept under $OUT/logs ($tag*)"
    [ $rc -gt $worst ] && worst=$rc
  else
    for f in $OUT/logs/${tag}_*.log; do [ -f "$f" ] && tail -3 "$f" > "$f.tail" && rm -f "$f"; done
  fi
}
run single_counts single --iters ${2:-160000} --procs 8 --counts --seed $S
run single_counts_reprojection reproj --iters ${3:-80000} --procs 8 --counts --modes 1,2 --seed $((S+10101))
run batches batch --batch --iters ${4:-6000} --procs 6 --seed $((S+20202))
exit $worst
