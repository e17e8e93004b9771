#!/bin/bash
# Soak of the round-6 library (seeds unlike any earlier soak): the pipelined stream with packed frames, small chunks and every push
# form; the batch queue in both submission forms; random batches through the staged scoring; single-pair configurations with every
# hypothesis's count compared (kernels 1 - 2 index frames through strides now).  Summaries in $OUT/*.txt.
OUT=${1:-gpurun_out/r06_soak}; mkdir -p $OUT/logs
S=${2:-606060}
N_STREAM=${3:-3000}; N_QUEUE=${4:-1500}; N_SINGLE=${5:-120000}; N_BATCH=${6:-4000}
export PYTHONPATH=$PWD:$PWD/tests
for w in 0 1 2 3 4 5; do (python3 tests/test_gpu_stream_async.py $N_STREAM $((S + w)) > $OUT/logs/stream_$w.log 2>&1; tail -1 $OUT/logs/stream_$w.log) & done; wait
for w in 0 1 2 3; do (python3 tests/test_gpu_batch_queue.py $N_QUEUE $((S + 100 + w)) > $OUT/logs/queue_$w.log 2>&1; tail -1 $OUT/logs/queue_$w.log) & done; wait
cat $OUT/logs/stream_*.log | grep "done\|MISMATCH" > $OUT/stream_fuzz.txt
cat $OUT/logs/queue_*.log | grep "done\|MISMATCH" > $OUT/queue_fuzz.txt
if [ "$N_SINGLE" -gt 0 ]; then python3 tests/fuzz_gpu.py --iters $N_SINGLE --procs 8 --counts --seed $((S + 1000)) --log-dir $OUT/logs --tag single > $OUT/single_counts.txt 2>&1; fi
if [ "$N_BATCH" -gt 0 ]; then python3 tests/fuzz_gpu.py --batch --iters $N_BATCH --procs 6 --seed $((S + 2000)) --log-dir $OUT/logs --tag batch > $OUT/batches.txt 2>&1; fi
tail -n 2 -q $OUT/stream_fuzz.txt $OUT/queue_fuzz.txt $OUT/single_counts.txt $OUT/batches.txt 2>/dev/null
for f in $OUT/logs/*.log; do tail -3 "$f" > "$f.tail"; rm -f "$f"; done
