#!/bin/bash
# Same box, alternating runs of several library builds: ab_lib3.sh <outdir> "<suffix list>" [bench args ...]   ("" = the current library)
out=$1; shift; sufs=$1; shift
mkdir -p $out
i=0
for rep in 1 2 3; do for v in $sufs; do
  [ "$v" = "cur" ] && v=""
  lib=$PWD/putslam_amd/libputslam_hip$v.so
  [ -f $lib ] || continue
  i=$((i+1))
  PUTSLAM_HIP_LIB=$lib python3 bench.py --streams 1 --steps 20 --warmup 20 --repeats 3 --no-cpu-baseline --no-other-modes "$@" > $out/run${i}${v:-_cur}.json 2>> $out/err.txt
done; done
python3 - <<PY
import json,glob
for f in sorted(glob.glob("$out/run*.json"), key=lambda x:int(x.split('run')[-1].split('_')[0])):
    d=json.loads(open(f).read().strip().splitlines()[-1])
    print(f.split('/')[-1].ljust(18), "ms/step=%.3f"%d["ms_per_step"], "k3=%.4f"%d["kernel_ms"]["ps_ransac_score"])
PY
