#!/bin/bash
# kernel-3 build variants of the Euclidean fast kernel (waves per SIMD, record prefetch) on E0 fixed/4096 and ransac/487
out=${1:-gpurun_out/r03a_var}
mkdir -p $out
for v in ""; do
  lib=$PWD/putslam_amd/libputslam_hip$v.so
  [ -f $lib ] || continue
  PUTSLAM_HIP_LIB=$lib python3 bench.py --streams 1 --steps 10 --warmup 3 --error-version 0 --no-cpu-baseline > $out/E0_fixed4096$v.json 2> $out/E0_fixed4096$v.err
  PUTSLAM_HIP_LIB=$lib python3 bench.py --streams 1 --steps 10 --warmup 3 --error-version 0 --estimator ransac --hyp 487 --no-cpu-baseline > $out/E0_ransac487$v.json 2> $out/E0_ransac487$v.err
done
python3 - <<PY
import json,glob
for f in sorted(glob.glob("$out/E0_*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split('/')[-1], "pairs/s=%.0f"%d["value"], "ms/step=%.3f"%d["ms_per_step"], {k:round(v,4) for k,v in d["kernel_ms"].items()})
    except Exception as e:
        print(f, "ERR", e)
PY
