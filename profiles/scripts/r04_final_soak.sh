#!/bin/bash
# Final soak of the round-4 library (different seeds from every earlier soak): single-pair configurations with EVERY hypothesis's
# count compared, all error versions / schedules, and random batches through the staged scoring (reordered / original order /
# complete / oracle).  Full worker logs under $OUT/logs; summaries in $OUT/*.txt.
OUT=${1:-gpurun_out/r04_final_soak}; mkdir -p $OUT
S=${5:-404040} # seed base (r04f: 404040; r04g, after the latency work: 505050)
python3 tests/fuzz_gpu.py --iters ${2:-160000} --procs 8 --counts --seed $S --log-dir $OUT/logs --tag single > $OUT/single_counts.txt 2>&1
python3 tests/fuzz_gpu.py --iters ${3:-80000} --procs 8 --counts --modes 1,2 --seed $((S+10101)) --log-dir $OUT/logs --tag reproj > $OUT/single_counts_reprojection.txt 2>&1
python3 tests/fuzz_gpu.py --batch --iters ${4:-6000} --procs 6 --seed $((S+20202)) --log-dir $OUT/logs --tag batch > $OUT/batches.txt 2>&1
tail -n 2 -q $OUT/single_counts.txt $OUT/single_counts_reprojection.txt $OUT/batches.txt
# keep only the tails of the per-worker logs (the full logs of a failing worker are printed by the harness itself)
for f in $OUT/logs/*.log; do tail -3 "$f" > "$f.tail"; rm -f "$f"; done
