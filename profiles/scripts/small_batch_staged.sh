#!/bin/bash
# Small batches through PsBatchQueue (4 chains) and through one context: the cost model's choice (complete scoring below its
# crossover) against the staged form forced (PUTSLAM_HIP_PRUNE=2).  Fixed schedule, H = 4096.  Output: gpurun_out/r06u/small_batch_staged.txt
out=gpurun_out/r06u; mkdir -p $out
f=$out/small_batch_staged.txt; : > $f
for ev in 1 0; do
  echo "## errorVersion $ev, fixed, H = 4096" >> $f
  for frames in 5 9 17 33 49 65 97; do
    steps=$(( 20000 / frames )); [ $steps -gt 400 ] && steps=400
    for chains in 1 4; do
      for prune in 1 2; do
        line=$(PUTSLAM_HIP_PRUNE=$prune timeout 120 ./demos/cpp/demo_batch_queue --frames $frames --chains $chains --error-version $ev --estimator fixed --hyp 4096 --steps $steps --warmup 5 --warm-seconds 0.4 --repeats 3 | tail -1)
        echo "frames $frames chains $chains prune $prune: $line" | cut -c1-110 >> $f
      done
    done
  done
done
cat $f
