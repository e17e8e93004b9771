#!/bin/bash
# kernel 1: work-group-side expansion through LDS (default) vs the FP4 image written by a launch of its own (round 2)
out=${1:-gpurun_out/r03g}
mkdir -p $out
for f in 1 0 1 0; do
  PUTSLAM_HIP_MATCHER_FUSED=$f python3 bench.py --streams 1 --steps 10 --warmup 5 --repeats 3 --error-version 0 --estimator ransac --hyp 487 --no-cpu-baseline --no-other-modes > $out/E0_ransac487_fused${f}_$RANDOM.json 2>> $out/err.txt
done
for f in 1 0; do
  PUTSLAM_HIP_MATCHER_FUSED=$f python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-other-modes > $out/default_fused$f.json 2>> $out/err.txt
done
python3 - <<PY
import json,glob
for f in sorted(glob.glob("$out/*.json")):
    d=json.loads(open(f).read().strip().splitlines()[-1])
    print(f.split('/')[-1], "pairs/s=%.0f"%d["value"], "ms/step=%.3f"%d["ms_per_step"], {k:round(v,4) for k,v in d["kernel_ms"].items()})
PY
