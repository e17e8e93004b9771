#!/bin/bash
# the one failing soak command of round 3 (gpurun_out/call_twins.log:21), repeated N times per
# library: the round-3 library from just BEFORE commit 32d8835 (no barrier after the list stages' meeting slots are
# cleared), round 3's final library, and the current one.  Every worker's complete output is kept (tests/fuzz_gpu.py
# --log-dir); the summary line of every repetition goes to $OUT/summary_<lib>.txt.
# (PUTSLAM_HIP_PRUNE=0 is kept for fidelity; run_batch sets prune per context itself, so it never had an effect there.)
#   usage: r04_soak_repro.sh <reps> <out dir> <lib tag>=<path to .so> ...
REPS=${1:-50}; OUT=${2:-gpurun_out/r04_soak}; shift 2
mkdir -p "$OUT"
for spec in "$@"; do
  tag=${spec%%=*}; lib=${spec#*=}
  : > "$OUT/summary_$tag.txt"
  for i in $(seq 1 "$REPS"); do
    PUTSLAM_HIP_LIB="$lib" PUTSLAM_HIP_PRUNE=0 python3 tests/fuzz_gpu.py --batch --iters 120 --procs 6 --seed 61200 \
        --log-dir "$OUT/logs_$tag" --tag "rep$i" > "$OUT/rep_${tag}_$i.txt" 2>&1
    rc=$?
    echo "rep $i rc $rc: $(tail -1 "$OUT/rep_${tag}_$i.txt")" >> "$OUT/summary_$tag.txt"
    if [ $rc -ne 0 ]; then cp "$OUT/rep_${tag}_$i.txt" "$OUT/FAILED_${tag}_$i.txt"; else rm -f "$OUT/rep_${tag}_$i.txt"; rm -rf "$OUT/logs_$tag"; fi
  done
  echo "== $tag: $(grep -c 'rc 0' "$OUT/summary_$tag.txt") of $REPS repetitions clean" | tee -a "$OUT/summary.txt"
done
