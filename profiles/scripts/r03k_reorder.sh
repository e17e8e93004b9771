#!/bin/bash
# Staged scoring with the reordered match record (PUTSLAM_HIP_REORDER=0/1): tests, then single-chain A/B of the bench regimes
out=${1:-gpurun_out/r03k}
mkdir -p $out
python3 -m pytest tests/test_gpu_prune.py tests/test_gpu_fuzz_slice.py tests/test_gpu_batch.py tests/test_gpu_score_euclid.py -x -q 2>&1 | tail -3
python3 tests/fuzz_gpu.py --batch --iters 160 --procs 4 --seed 700 2>&1 | tail -3
for ro in 1 0 1 0; do
  for ev in 0 1; do
    PUTSLAM_HIP_REORDER=$ro python3 bench.py --streams 1 --steps 10 --warmup 10 --repeats 3 --error-version $ev --no-cpu-baseline --no-other-modes > $out/E${ev}_fixed4096_reorder${ro}_$RANDOM.json 2>> $out/err.txt
  done
done
PUTSLAM_HIP_REORDER=1 python3 bench.py --streams 1 --steps 10 --warmup 10 --repeats 3 --error-version 2 --no-cpu-baseline --no-other-modes > $out/E2_fixed4096_reorder1.json 2>> $out/err.txt
PUTSLAM_HIP_REORDER=0 python3 bench.py --streams 1 --steps 10 --warmup 10 --repeats 3 --error-version 2 --no-cpu-baseline --no-other-modes > $out/E2_fixed4096_reorder0.json 2>> $out/err.txt
python3 - <<PY
import json,glob
for f in sorted(glob.glob("$out/E*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split('/')[-1], "pairs/s=%.0f"%d["value"], "ms/step=%.3f"%d["ms_per_step"], "evals=%.3f"%d.get("score_evals_frac",-1), {k:round(v,4) for k,v in d["kernel_ms"].items()})
    except Exception as e:
        print(f, "ERR", e)
PY
