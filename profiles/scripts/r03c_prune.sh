#!/bin/bash
# pruned scoring A/B (PUTSLAM_HIP_PRUNE=0/1), single chain, the regimes of the bench line
out=${1:-gpurun_out/r03c}
mkdir -p $out
for pr in 1 0; do
  for ev in 0 1; do
    PUTSLAM_HIP_PRUNE=$pr python3 bench.py --streams 1 --steps 10 --warmup 3 --repeats 3 --error-version $ev --no-cpu-baseline --no-other-modes > $out/E${ev}_fixed4096_prune$pr.json 2>> $out/err.txt
    PUTSLAM_HIP_PRUNE=$pr python3 bench.py --streams 1 --steps 10 --warmup 3 --repeats 3 --error-version $ev --estimator ransac --hyp 487 --no-cpu-baseline --no-other-modes > $out/E${ev}_ransac487_prune$pr.json 2>> $out/err.txt
  done
  PUTSLAM_HIP_PRUNE=$pr python3 bench.py --preset stress --streams 1 --steps 5 --warmup 2 --repeats 3 --error-version 0 --no-cpu-baseline --no-other-modes > $out/E0_stress_prune$pr.json 2>> $out/err.txt
done
python3 - <<PY
import json,glob
for f in sorted(glob.glob("$out/E*_prune*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split('/')[-1], "pairs/s=%.0f"%d["value"], "ms/step=%.3f"%d["ms_per_step"], {k:round(v,4) for k,v in d["kernel_ms"].items()})
    except Exception as e:
        print(f, "ERR", e)
PY
