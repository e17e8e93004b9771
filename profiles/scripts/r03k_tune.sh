#!/bin/bash
# Tuning sweep of the reordered staged scoring: voters (TOP), end of stage 1 (MARGIN), end of stage 2 (C2DIV)
out=${1:-gpurun_out/r03k_tune}
mkdir -p $out
python3 -m pytest tests/test_gpu_prune.py tests/test_gpu_fuzz_slice.py -x -q 2>&1 | tail -2
run() { # name env...
  name=$1; shift
  for ev in 1 0; do
    env "$@" python3 bench.py --streams 1 --steps 8 --warmup 8 --repeats 2 --error-version $ev --no-cpu-baseline --no-other-modes > $out/E${ev}_$name.json 2>> $out/err.txt
  done
}
run off PUTSLAM_HIP_REORDER=0
for top in 1 4 8 16; do run top${top}_m48_d4 PUTSLAM_HIP_REORDER_TOP=$top; done
for m in 1 16 112 176; do run top8_m${m}_d4 PUTSLAM_HIP_REORDER_MARGIN=$m; done
for d in 2 3 8; do run top8_m48_d$d PUTSLAM_HIP_REORDER_C2DIV=$d; done
run top16_m16_d8 PUTSLAM_HIP_REORDER_TOP=16 PUTSLAM_HIP_REORDER_MARGIN=16 PUTSLAM_HIP_REORDER_C2DIV=8
for ro in 0 1; do
 PUTSLAM_HIP_REORDER=$ro python3 bench.py --streams 1 --steps 10 --warmup 10 --repeats 3 --error-version 0 --estimator ransac --hyp 487 --no-cpu-baseline --no-other-modes > $out/R0_ransac487_reorder$ro.json 2>> $out/err.txt
done
python3 - <<PY
import json,glob
for f in sorted(glob.glob("$out/*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split('/')[-1].ljust(28), "ms/step=%.3f"%d["ms_per_step"], "evals=%.3f"%(d.get("score_evals_frac") or -1), "k3=%.4f"%d["kernel_ms"]["ps_ransac_score"], "sum=%.4f"%d["single_chain"]["kernel_ms_sum"])
    except Exception as e:
        print(f, "ERR", e)
PY
