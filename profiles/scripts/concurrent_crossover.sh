#!/bin/bash
# Where the staged scoring starts to pay when batches run side by side (PsBatchQueue, 4 chains): the cost model's choice
# (PUTSLAM_HIP_PRUNE=1) against complete scoring (0) and the staged form forced (2), over keypoints x H x batch size.
# Output: gpurun_out/r06u/concurrent_crossover.txt -- "ev est H kpts pairs chains prune pairs_per_s"
out=gpurun_out/r06u; mkdir -p $out
f=$out/concurrent_crossover.txt; : > $f
chains=${CHAINS:-4}
run() { # ev est hyp kpts frames
  for prune in 0 2; do
    steps=$(( 6000000 / ($4 * $5) )); [ $steps -gt 300 ] && steps=300; [ $steps -lt 20 ] && steps=20
    r=$(PUTSLAM_HIP_PRUNE=$prune timeout 120 ./demos/cpp/demo_batch_queue --frames $5 --kpts $4 --chains $chains --error-version $1 --estimator $2 --hyp $3 --steps $steps --warmup 3 --warm-seconds 0.15 --repeats 3 | tail -1 | sed -e 's/.*median \([0-9]*\) .*/\1/')
    echo "$1 $2 $3 $4 $(( $5 - 1 )) $chains $prune $r" >> $f
  done
}
for ev in 1 0; do
  for kpts in 500 2000 4000; do
    for hyp in 1024 4096 16384; do
      for frames in 3 5 9 17 33 65; do run $ev fixed $hyp $kpts $frames; done
    done
  done
  for est in ransac usac; do
    for frames in 3 5 9 17 33 65; do run $ev $est $([ $est = ransac ] && echo 487 || echo 850000) 2000 $frames; done
  done
done
cat $f
