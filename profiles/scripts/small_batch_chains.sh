#!/bin/bash
# Batches through PsBatchQueue: batch size x chains (demos/cpp/demo_batch_queue, synthetic frames, E1 / fixed / H = 4096 and
# the reference's own regime E0 / RANSAC <= 487).  Output: gpurun_out/r06u/small_batch_chains.txt
#   small_batch_chains.sh ["17 33 65 126 251"] ["1 2 3 4"]
out=gpurun_out/r06u; mkdir -p $out
f=$out/small_batch_chains.txt
sizes=${1:-"17 33 65 126 251 500 1001"}
chainset=${2:-"1 2 3 4 6 8"}
for ev in "1 fixed" "0 ransac"; do
  set -- $ev
  echo "## errorVersion $1, estimator $2" >> $f
  for frames in $sizes; do
    steps=$(( 20000 / frames )); [ $steps -gt 400 ] && steps=400
    for chains in $chainset; do
      line=$(timeout 120 ./demos/cpp/demo_batch_queue --frames $frames --chains $chains --error-version $1 --estimator $2 --hyp $([ $2 = fixed ] && echo 4096 || echo 487) --steps $steps --warmup 5 --warm-seconds 0.5 --repeats 3 | tail -1)
      echo "frames $frames chains $chains: $line" | cut -c1-110 >> $f
    done
  done
done
cat $f
