#!/bin/bash
# same box, same process conditions: round-2 library vs the current one on the headline configuration, single chain
out=${1:-gpurun_out/r03b}
mkdir -p $out
for v in _r02 "" _r02 ""; do
  lib=$PWD/putslam_amd/libputslam_hip$v.so
  [ -f $lib ] || continue
  PUTSLAM_HIP_LIB=$lib python3 bench.py --streams 1 --steps 20 --warmup 20 --repeats 3 --no-cpu-baseline --no-other-modes > $out/E1_fixed4096${v}_$RANDOM.json 2>> $out/err.txt
done
python3 - <<PY
import json,glob
for f in sorted(glob.glob("$out/E1_*.json")):
    d=json.loads(open(f).read().strip().splitlines()[-1])
    print(f.split('/')[-1], "pairs/s=%.0f"%d["value"], "ms/step=%.3f"%d["ms_per_step"], {k:round(v,4) for k,v in d["kernel_ms"].items()})
PY
