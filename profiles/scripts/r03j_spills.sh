#!/bin/bash
# Kernel 3 without prologue spills (LDS parking, three stage kinds): tests, timing of the bench regimes, PMC traffic.
out=${1:-gpurun_out/r03j}
mkdir -p $out
python3 -m pytest tests/test_gpu_prune.py tests/test_gpu_score_variants.py tests/test_gpu_score_euclid.py tests/test_gpu_band_edges.py tests/test_gpu_batch.py -x -q 2>&1 | tail -3
for ev in 0 1 2; do
  python3 bench.py --streams 1 --steps 10 --warmup 10 --repeats 3 --error-version $ev --no-cpu-baseline --no-other-modes > $out/E${ev}_fixed4096.json 2>> $out/err.txt
done
python3 bench.py --streams 1 --steps 10 --warmup 10 --repeats 3 --error-version 0 --estimator ransac --hyp 487 --no-cpu-baseline --no-other-modes > $out/E0_ransac487.json 2>> $out/err.txt
python3 - <<PY
import json,glob
for f in sorted(glob.glob("$out/E*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split('/')[-1], "pairs/s=%.0f"%d["value"], "ms/step=%.3f"%d["ms_per_step"], {k:round(v,4) for k,v in d["kernel_ms"].items()})
    except Exception as e:
        print(f, "ERR", e)
PY
