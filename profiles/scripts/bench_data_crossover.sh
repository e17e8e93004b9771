#!/bin/bash
# The side-by-side crossover on bench.py's own kind of data (putslam_amd/synth.py, 70 % true correspondences: fewer inliers than the
# C++ demo's synthetic frames, so the staged form abandons less): PsBatchQueue, 4 chains, complete (0) / cost model (1) / staged (2).
# Output: gpurun_out/r06u/bench_data_crossover.txt
out=gpurun_out/r06u; mkdir -p $out /tmp/psq
f=$out/bench_data_crossover${TAG:-}.txt; : > $f
SIZES=${SIZES:-"5 9 17 33 65"}; CHAINS=${CHAINS:-4}
for frames in $SIZES; do
python3 - $frames <<'P'
import sys, numpy as np
sys.path.insert(0, '.')
from putslam_amd import synth
F = int(sys.argv[1])
for frac in (0.7, 0.4):
    seq = synth.make_sequence(F, 2000, config=3, index=0, inlier_frac=frac)
    cap = seq["desc"].shape[1]
    with open("/tmp/psq/seq_%d_%d.bin" % (F, int(frac * 100)), "wb") as f:
        np.array([F, cap], np.int32).tofile(f)
        np.ascontiguousarray(seq["nkpts"], np.int32).tofile(f)
        np.ascontiguousarray(seq["desc"], np.uint8).tofile(f)
        np.ascontiguousarray(seq["pts"], np.float32).tofile(f)
P
done
for frac in 70 40; do
for ev in "1 fixed 4096" "0 fixed 4096" "0 ransac 487" "1 ransac 487"; do
  set -- $ev
  for frames in $SIZES; do
    steps=$(( 6000 / frames )); [ $steps -gt 300 ] && steps=300
    line="chains $CHAINS inliers $frac E$1 $2 $3 pairs $(( frames - 1 )):"
    for prune in 0 1 2; do
      r=$(PUTSLAM_HIP_PRUNE=$prune timeout 120 ./demos/cpp/demo_batch_queue --sequence /tmp/psq/seq_${frames}_$frac.bin --chains $CHAINS --error-version $1 --estimator $2 --hyp $3 --steps $steps --warmup 3 --warm-seconds 0.2 --repeats 3 | tail -1 | sed -e 's/.*median \([0-9]*\) .*/\1/')
      line="$line  prune$prune $r"
    done
    echo "$line" >> $f
  done
done
done
cat $f
