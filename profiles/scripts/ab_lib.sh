#!/bin/bash
# Same box, alternating runs: libputslam_hip_prev.so (the previous commit, built beside the tree) against the current
# library, single chain.  usage: ab_lib.sh <outdir> [bench args ...]
out=${1:-gpurun_out/ab}; shift
mkdir -p $out
i=0
for v in _prev "" _prev "" _prev ""; do
  lib=$PWD/putslam_amd/libputslam_hip$v.so
  [ -f $lib ] || continue
  i=$((i+1))
  PUTSLAM_HIP_LIB=$lib python3 bench.py --streams 1 --steps 20 --warmup 20 --repeats 3 --no-cpu-baseline --no-other-modes "$@" > $out/run${i}${v:-_new}.json 2>> $out/err.txt
done
python3 - <<PY
import json,glob
for f in sorted(glob.glob("$out/run*.json")):
    d=json.loads(open(f).read().strip().splitlines()[-1])
    print(f.split('/')[-1], "pairs/s=%.0f"%d["value"], "ms/step=%.3f"%d["ms_per_step"], {k:round(v,4) for k,v in d["kernel_ms"].items()})
PY
