#!/bin/bash
# kernel 1 (fused form): fewer vector instructions -- block-scaled MFMA + add/max epilogue, LDS table for the expansion
out=${1:-gpurun_out/r03h}
mkdir -p $out
for rep in 1 2; do
for v in "" _tt6 _tt8; do
  lib=$PWD/putslam_amd/libputslam_hip$v.so
  [ -f $lib ] || continue
  PUTSLAM_HIP_LIB=$lib python3 bench.py --streams 1 --steps 10 --warmup 5 --repeats 3 --error-version 0 --estimator ransac --hyp 487 --no-cpu-baseline --no-other-modes > $out/E0_ransac487${v}_$rep.json 2>> $out/err.txt
done
done
python3 - <<PY
import json,glob
for f in sorted(glob.glob("$out/E0_ransac487*.json")):
    d=json.loads(open(f).read().strip().splitlines()[-1])
    print(f.split('/')[-1], "pairs/s=%.0f"%d["value"], "ms/step=%.3f"%d["ms_per_step"], {k:round(v,4) for k,v in d["kernel_ms"].items()}, round(d["kernel_bounds"]["ps_hamming_mfma"]["frac"],3))
PY
