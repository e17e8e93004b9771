#!/bin/bash
# A/B of kernel 3 for errorVersion 0: decision-exact packed kernel (default) vs value-exact ps_ransac_score<0>
# usage (GPU box): bash profiles/scripts/r03a_euclid_ab.sh <outdir>
out=${1:-gpurun_out/r03a}
mkdir -p $out
for score in fast exact; do
  PUTSLAM_HIP_SCORE=$score python3 bench.py --streams 1 --steps 10 --warmup 3 --error-version 0 --no-cpu-baseline > $out/bench_E0_fixed4096_$score.json 2> $out/bench_E0_fixed4096_$score.err
  PUTSLAM_HIP_SCORE=$score python3 bench.py --streams 1 --steps 10 --warmup 3 --error-version 0 --estimator ransac --hyp 487 --no-cpu-baseline > $out/bench_E0_ransac487_$score.json 2> $out/bench_E0_ransac487_$score.err
  PUTSLAM_HIP_SCORE=$score python3 bench.py --preset stress --streams 1 --steps 5 --warmup 2 --error-version 0 --no-cpu-baseline > $out/bench_E0_stress_$score.json 2> $out/bench_E0_stress_$score.err
done
python3 - <<PY
import json,glob
for f in sorted(glob.glob("$out/bench_E0_*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split('/')[-1], "pairs/s=%.0f"%d["value"], "ms/step=%.3f"%d["ms_per_step"], {k:round(v,4) for k,v in d["kernel_ms"].items()})
    except Exception as e:
        print(f, "ERR", e)
PY
